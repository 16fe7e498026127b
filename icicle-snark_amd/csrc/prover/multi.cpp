// multi.cpp — one Groth16 prove over a GROUP of devices inside ONE process: what groth16_prove() does for a device string
// such as "HIP:0-7" (the reference's entry point takes a device type and uses id 0: src/lib.rs:25-61, src/main.rs:46-70).
//
// Partition (HISTORY.md §5): shard r of G holds the point range r of the A, B1, B2, C bases, the residue class
// k ≡ r (mod G) of the H bases and computes 1/G of the QAP front end.  One host thread per shard enqueues that shard's
// whole pipeline on its own device; the shards meet in three device-side exchanges and once on the host:
//   1. witness       every shard uploads 1/G over PCIe; an in-place all-gather over xGMI completes the buffer on every device;
//   2. all-to-all 1  between stage 1 (spmv of the rows ≡ r, size-n/G inverse transform) and stage 2 (size-G DFTs, coset, twist);
//   3. all-to-all 2  between stage 2 and the size-n/G forward transform with the A·B − C epilogue;
//   4. the five partial commitments of every shard (Horner tails on host threads, as on one GPU) are summed on the host.
// The exchanges are ENQUEUED on each shard's QAP stream and ordered against the producers on the other devices by events —
// no host wait sits between the stages.  Transport, in the order tried (ICICLE_SNARK_EXCHANGE=pull|memcpy|rccl forces one):
//   pull    one kernel per rank and exchange that reads the peers' buffers directly (peer access over xGMI: all seven links of
//           the point-to-point fabric at once; the default, and plain device copies when several shards share a device);
//   memcpy  hipMemcpyPeerAsync per chunk (no peer mapping needed);
//   rccl    ncclAllGather / grouped ncclSend+ncclRecv on communicators of ncclCommInitAll (libicicle_snark_rccl.so, loaded on
//           demand so that the single-GPU library carries no RCCL dependency).
// Every transport is checked once per group with a data pattern before a prove relies on it.
#include <algorithm>
#include <dlfcn.h>

#include "../team.h"
#include "exchange.h"
#include "prover_internal.h"

using namespace bn254;
using namespace isnark;

namespace isnark {
namespace prover {

int set_active_device(int device_id); // prover.cpp

// ------------------------------------------------------------------------------------------------ device string
static bool parse_id_list(const char* p, std::vector<int>& ids)
{
  // <item>{,<item>}   item = <id> | <first>-<last>
  while (*p) {
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p || a < 0 || a > 1023) return false;
    long b = a;
    p = end;
    if (*p == '-') {
      p++;
      b = strtol(p, &end, 10);
      if (end == p || b < a || b > 1023) return false;
      p = end;
    }
    for (long k = a; k <= b; k++) ids.push_back((int)k);
    if (*p == ',') {
      p++;
      if (!*p) return false;
    } else if (*p)
      return false;
  }
  return !ids.empty();
}

int parse_device_string(const char* device, std::vector<int>& ids)
{
  ids.clear();
  if (!device) return fail(ERR_ARG, "null device string");
  const char* colon = strchr(device, ':');
  const std::string type = colon ? std::string(device, colon) : std::string(device);
  if (type != "HIP" && type != "CUDA")
    return fail((int)ICICLE_INVALID_DEVICE, "device type '%.63s' is not registered (only HIP; this library has no CPU fallback)", type.c_str());
  const char* list = colon ? colon + 1 : getenv("ICICLE_SNARK_DEVICES");
  if (!colon && (!list || !*list)) {
    ids.push_back(0); // Device::new(device, 0) — src/lib.rs:26
    return 0;
  }
  if (!parse_id_list(list, ids) || ids.size() > XCHG_MAX_PEERS) {
    ids.clear();
    return fail((int)ICICLE_INVALID_DEVICE, "bad device list '%.200s' (expected e.g. HIP:0-7 or HIP:0,2,4; at most %u shards)", list, XCHG_MAX_PEERS);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ exchange
struct RcclApi {
  void* dso = nullptr;
  int (*init_all)(int, const int*, void**) = nullptr;
  int (*allgather_on)(void*, void*, size_t, void*) = nullptr;
  int (*alltoall_on)(void*, const void*, void*, int, size_t, size_t, void*) = nullptr;
  int (*destroy)(void*) = nullptr;
  const char* (*last_error)(void) = nullptr;
};

enum ExchangeMode { XCHG_PULL = 0, XCHG_MEMCPY = 1, XCHG_RCCL = 2 };
static const char* mode_name(int m) { return m == XCHG_PULL ? "pull" : m == XCHG_MEMCPY ? "memcpy" : "rccl"; }

struct DeviceGroup {
  std::vector<int> devs;
  std::vector<std::shared_ptr<ZKeyCache>> shards;
  std::unique_ptr<Team> team;
  bool dist = false; // every shard can run the distributed front end
  bool peer_ok = false; // every device of the group maps every other one's memory (or they are one device)
  int mode = XCHG_PULL;
  RcclApi rccl;
  std::vector<void*> comms;
  std::vector<hipEvent_t> ev_slice, ev_s1, ev_s2; // per shard, recorded on its s_qap: slice resident / stage 1 done / stage 2 done
  ~DeviceGroup()
  {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    team.reset();
    for (size_t r = 0; r < comms.size(); r++)
      if (comms[r] && rccl.destroy) (void)rccl.destroy(comms[r]);
    for (size_t r = 0; r < shards.size(); r++) {
      (void)hipSetDevice(devs[r]);
      for (auto* v : {&ev_slice, &ev_s1, &ev_s2})
        if (r < v->size() && (*v)[r]) (void)hipEventDestroy((*v)[r]);
    }
    shards.clear();
    if (rccl.dso) dlclose(rccl.dso);
  }
};

const ZKeyCache* group_lead(const DeviceGroup* g) { return g->shards[0].get(); }
void group_info(const DeviceGroup* g, Groth16CircuitInfo* info)
{
  const ZKeyCache* z = group_lead(g);
  info->n_vars = z->n_vars;
  info->n_public = z->n_public;
  info->domain_size = z->domain_size;
  info->n_coef = z->n_coef;
  info->device_bytes = 0;
  info->b_bases = 0;
  for (auto& s : g->shards) {
    info->device_bytes += s->device_bytes;
    info->b_bases += s->B1.len();
  }
  info->shards = (uint32_t)g->shards.size();
}

// what the group runs on, as one line of JSON (groth16_group_describe; bench.py reports it: the reader of a scaling run must
// be able to tell which transport moved the exchanges and whether RCCL saw the devices at all)
std::string group_describe(const DeviceGroup* g)
{
  std::vector<int> distinct;
  for (int d : g->devs)
    if (std::find(distinct.begin(), distinct.end(), d) == distinct.end()) distinct.push_back(d);
  std::string out = "{\"shards\": " + std::to_string(g->devs.size()) + ", \"devices\": [";
  for (size_t i = 0; i < g->devs.size(); i++) out += (i ? ", " : "") + std::to_string(g->devs[i]);
  out += "], \"distinct_devices\": " + std::to_string(distinct.size());
  out += std::string(", \"transport\": \"") + mode_name(g->mode) + "\"";
  out += std::string(", \"peer_access\": ") + (g->peer_ok ? "true" : "false");
  // ranks of an RCCL communicator that take part in the exchanges: 0 unless the rccl transport was selected
  out += ", \"rccl_ranks\": " + std::to_string(g->mode == XCHG_RCCL ? g->comms.size() : (size_t)0);
  out += std::string(", \"distributed_front_end\": ") + (g->dist ? "true" : "false");
  out += std::string(", \"transport_forced_by_env\": ") + (getenv("ICICLE_SNARK_EXCHANGE") && *getenv("ICICLE_SNARK_EXCHANGE") ? "true" : "false");
  out += "}";
  return out;
}

static bool load_rccl(RcclApi& api)
{
  if (api.dso) return true;
  // libicicle_snark_rccl.so sits next to this library
  Dl_info di;
  std::string path = "libicicle_snark_rccl.so";
  if (dladdr((const void*)&load_rccl, &di) && di.dli_fname) {
    std::string self = di.dli_fname;
    const size_t slash = self.rfind('/');
    if (slash != std::string::npos) path = self.substr(0, slash + 1) + path;
  }
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fail((int)ICICLE_BACKEND_LOAD_FAILED, "cannot load %s: %s", path.c_str(), dlerror());
    return false;
  }
  api.init_all = (int (*)(int, const int*, void**))dlsym(h, "icicle_snark_rccl_init_all");
  api.allgather_on = (int (*)(void*, void*, size_t, void*))dlsym(h, "icicle_snark_rccl_allgather_device_on");
  api.alltoall_on = (int (*)(void*, const void*, void*, int, size_t, size_t, void*))dlsym(h, "icicle_snark_rccl_alltoall_rows_on");
  api.destroy = (int (*)(void*))dlsym(h, "icicle_snark_rccl_destroy");
  api.last_error = (const char* (*)(void))dlsym(h, "icicle_snark_rccl_last_error");
  if (!api.init_all || !api.allgather_on || !api.alltoall_on || !api.destroy || !api.last_error) {
    dlclose(h);
    fail((int)ICICLE_BACKEND_LOAD_FAILED, "%s lacks the stream-ordered entry points", path.c_str());
    return false;
  }
  api.dso = h;
  return true;
}

// rank r's in-place all-gather of bufs[·] (G slices of slice_bytes each) on stream s; `ready[p]` = rank p's slice is in place
static int xchg_allgather(DeviceGroup* g, int r, void* const* bufs, size_t slice_bytes, const hipEvent_t* ready, hipStream_t s)
{
  const int G = (int)g->devs.size();
  if (g->mode == XCHG_RCCL) {
    if (g->rccl.allgather_on(g->comms[r], bufs[r], slice_bytes, s)) return fail((int)ICICLE_UNKNOWN_ERROR, "rccl all-gather: %s", g->rccl.last_error());
    return 0;
  }
  for (int p = 0; p < G; p++)
    if (p != r) P_HIP(hipStreamWaitEvent(s, ready[p], 0));
  if (g->mode == XCHG_PULL) {
    PeerPtrs pp;
    memset(&pp, 0, sizeof pp);
    for (int p = 0; p < G; p++) pp.p[p] = bufs[p];
    P_HIP(xchg_allgather_pull(pp, (uint32_t)G, (uint32_t)r, slice_bytes, s));
    return 0;
  }
  for (int p = 0; p < G; p++) {
    if (p == r) continue;
    const size_t off = (size_t)p * slice_bytes;
    P_HIP(hipMemcpyPeerAsync((uint8_t*)bufs[r] + off, g->devs[r], (const uint8_t*)bufs[p] + off, g->devs[p], slice_bytes, s));
  }
  return 0;
}

// rank r's side of an all-to-all of [row][peer][chunk] buffers: recv[q][p] ← sends[p][q][r]; `ready[p]` = sends[p] is complete
static int xchg_alltoall(DeviceGroup* g, int r, void* const* sends, void* recv, uint32_t rows, size_t row_bytes, size_t chunk_bytes, const hipEvent_t* ready, hipStream_t s)
{
  const int G = (int)g->devs.size();
  if (g->mode == XCHG_RCCL) {
    if (g->rccl.alltoall_on(g->comms[r], sends[r], recv, (int)rows, row_bytes, chunk_bytes, s)) return fail((int)ICICLE_UNKNOWN_ERROR, "rccl all-to-all: %s", g->rccl.last_error());
    return 0;
  }
  for (int p = 0; p < G; p++)
    if (p != r) P_HIP(hipStreamWaitEvent(s, ready[p], 0));
  if (g->mode == XCHG_PULL) {
    PeerPtrs pp;
    memset(&pp, 0, sizeof pp);
    for (int p = 0; p < G; p++) pp.p[p] = sends[p];
    P_HIP(xchg_alltoall_pull(pp, recv, (uint32_t)G, (uint32_t)r, rows, row_bytes, chunk_bytes, s));
    return 0;
  }
  for (uint32_t q = 0; q < rows; q++)
    for (int p = 0; p < G; p++) {
      const uint8_t* src = (const uint8_t*)sends[p] + (size_t)q * row_bytes + (size_t)r * chunk_bytes;
      uint8_t* dst = (uint8_t*)recv + (size_t)q * row_bytes + (size_t)p * chunk_bytes;
      P_HIP(hipMemcpyPeerAsync(dst, g->devs[r], src, g->devs[p], chunk_bytes, s));
    }
  return 0;
}

// ---- failure bookkeeping of one team run: the first error wins, every shard keeps walking through the barriers
struct RunState {
  std::mutex mu;
  int rc = 0;
  std::string text;
  std::atomic<bool> failed{false};
  void note(int code)
  {
    if (!code) return;
    std::lock_guard<std::mutex> lk(mu);
    if (!rc) {
      rc = code;
      text = last_error_text();
    }
    failed.store(true);
  }
  int finish() const
  {
    if (rc) set_error_text(text.c_str());
    return rc;
  }
};

// Pattern all-gather + all-to-all through the group's transport, verified on the host — TWO rounds with different patterns on
// the SAME buffers and the SAME events, which is what a sequence of proves does to d_witness / d_dist_y / d_dist_send2 and
// ev_slice / ev_s1 / ev_s2.  The producers are made LATE on purpose in both rounds — a delay kernel (longer on every next rank)
// sits in front of the copies that put the patterns in place, and nothing synchronises the host before the peers enqueue their
// side — so a transport whose cross-device ordering does not hold (an event of one device that fails to hold back a stream of
// another) reads the zeroed buffers in round 1, and one that keeps stale remote lines on the reader or does not honour the
// RE-record of an event reads round 1's bytes in round 2: either is rejected here, not in a proof.
static int exchange_self_test(DeviceGroup* g)
{
  const int G = (int)g->devs.size();
  const size_t slice = 4096, chunk = 1024, rows = 3, row_bytes = chunk * G;
  const int ROUNDS = 2;
  std::vector<uint8_t*> ag(G, nullptr), snd(G, nullptr), rcv(G, nullptr), stage(G, nullptr);
  std::vector<void*> agv(G), sndv(G);
  RunState st;
  std::vector<int> bad(G, 0);
  auto ag_byte = [](int round, int rank, size_t i) { return (uint8_t)(rank * 31 + i * 7 + 1 + round * 101); };
  auto a2a_byte = [](int round, int src, int dst, size_t q, size_t i) { return (uint8_t)(src * 17 + dst * 5 + q * 3 + i + round * 59); };
  g->team->run([&](int r) {
    int rc = set_active_device(g->devs[r]);
    ZKeyCache* z = g->shards[r].get();
    hipStream_t s = z->s_qap;
    // exchange buffers zeroed — complete before anybody goes on
    auto prepare = [&]() -> int {
      P_HIP(hipMalloc((void**)&ag[r], slice * G));
      P_HIP(hipMalloc((void**)&snd[r], rows * row_bytes));
      P_HIP(hipMalloc((void**)&rcv[r], rows * row_bytes));
      P_HIP(hipMalloc((void**)&stage[r], slice + rows * row_bytes));
      P_HIP(hipMemsetAsync(ag[r], 0, slice * G, s));
      P_HIP(hipMemsetAsync(snd[r], 0, rows * row_bytes, s));
      P_HIP(hipMemsetAsync(rcv[r], 0, rows * row_bytes, s));
      P_HIP(hipStreamSynchronize(s));
      return 0;
    };
    // this round's patterns into the staging buffer (synchronised: the LATE part is the device-side copy below)
    auto fill_stage = [&](int round) -> int {
      std::vector<uint8_t> h(slice + rows * row_bytes, 0);
      for (size_t i = 0; i < slice; i++) h[i] = ag_byte(round, r, i);
      for (size_t q = 0; q < rows; q++)
        for (int p = 0; p < G; p++)
          for (size_t i = 0; i < chunk; i++) h[slice + q * row_bytes + (size_t)p * chunk + i] = a2a_byte(round, r, p, q, i);
      P_HIP(hipMemcpyAsync(stage[r], h.data(), h.size(), hipMemcpyHostToDevice, s));
      P_HIP(hipStreamSynchronize(s));
      return 0;
    };
    // the late producer: delay, then the patterns move into the buffers the peers read; the event is all the peers get.
    // The delays differ by rank (this rank's own exchange sits behind its own delay on the same stream): rank 0 is through
    // after 2 ms and would read rank 1's buffers 2 ms too early, rank 1 rank 2's, …
    auto produce = [&]() -> int {
      P_HIP(xchg_delay(2.0 * (r + 1), s));
      P_HIP(hipMemcpyAsync(ag[r] + (size_t)r * slice, stage[r], slice, hipMemcpyDeviceToDevice, s));
      P_HIP(hipMemcpyAsync(snd[r], stage[r] + slice, rows * row_bytes, hipMemcpyDeviceToDevice, s));
      P_HIP(hipEventRecord(g->ev_slice[r], s));
      return 0;
    };
    if (!rc) rc = prepare();
    st.note(rc);
    agv[r] = ag[r];
    sndv[r] = snd[r];
    for (int round = 0; round < ROUNDS; round++) {
      if (!st.failed) {
        rc = fill_stage(round);
        st.note(rc);
      }
      g->team->barrier(); // every buffer exists (round 0: zeroed); nobody still reads what the producers are about to overwrite
      if (!st.failed) {
        rc = produce();
        st.note(rc);
      }
      g->team->barrier(); // every ev_slice is recorded (enqueued, not complete)
      if (!st.failed) {
        rc = xchg_allgather(g, r, agv.data(), slice, g->ev_slice.data(), s);
        if (!rc) rc = xchg_alltoall(g, r, sndv.data(), rcv[r], (uint32_t)rows, row_bytes, chunk, g->ev_slice.data(), s);
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail((int)ICICLE_UNKNOWN_ERROR, "exchange self-test: stream error");
        st.note(rc);
      }
      if (!st.failed) {
        std::vector<uint8_t> a(slice * G), b(rows * row_bytes);
        (void)hipMemcpy(a.data(), ag[r], a.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), rcv[r], b.size(), hipMemcpyDeviceToHost);
        for (int p = 0; p < G && !bad[r]; p++)
          for (size_t i = 0; i < slice; i++)
            if (a[(size_t)p * slice + i] != ag_byte(round, p, i)) {
              bad[r] = 1 + round;
              break;
            }
        for (size_t q = 0; q < rows && !bad[r]; q++)
          for (int p = 0; p < G && !bad[r]; p++)
            for (size_t i = 0; i < chunk; i++)
              if (b[q * row_bytes + (size_t)p * chunk + i] != a2a_byte(round, p, r, q, i)) {
                bad[r] = 1 + round;
                break;
              }
      }
    }
    g->team->barrier(); // nobody frees a buffer a peer may still be reading
    for (uint8_t* p : {ag[r], snd[r], rcv[r], stage[r]})
      if (p) (void)hipFree(p);
  });
  if (int rc = st.finish()) return rc;
  for (int r = 0; r < G; r++)
    if (bad[r]) return fail((int)ICICLE_UNKNOWN_ERROR, "exchange self-test (%s): rank %d received wrong data in round %d", mode_name(g->mode), r, bad[r]);
  return 0;
}

// peer mappings for the pull kernels; called BEFORE the shard caches are allocated, so that every buffer a peer will read
// is created with the mapping in place (and the self-test below exercises the same kind of allocation)
static bool enable_peer_access(const std::vector<int>& devs)
{
  const int G = (int)devs.size();
  for (int i = 0; i < G; i++) {
    if (hipSetDevice(devs[i]) != hipSuccess) return false;
    for (int j = 0; j < G; j++) {
      if (devs[i] == devs[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devs[i], devs[j]) != hipSuccess || !can) {
        (void)hipGetLastError();
        return false;
      }
      const hipError_t e = hipDeviceEnablePeerAccess(devs[j], 0);
      (void)hipGetLastError();
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return false;
    }
  }
  return true;
}

static int setup_exchange(DeviceGroup* g)
{
  const int G = (int)g->devs.size();
  bool distinct = true;
  const bool peer_ok = g->peer_ok;
  for (int i = 0; i < G; i++)
    for (int j = 0; j < i; j++)
      if (g->devs[i] == g->devs[j]) distinct = false;
  std::vector<int> order;
  const char* forced = getenv("ICICLE_SNARK_EXCHANGE");
  if (forced && *forced) {
    if (!strcmp(forced, "pull")) order = {XCHG_PULL};
    else if (!strcmp(forced, "memcpy")) order = {XCHG_MEMCPY};
    else if (!strcmp(forced, "rccl")) order = {XCHG_RCCL};
    else return fail(ERR_ARG, "ICICLE_SNARK_EXCHANGE=%s: expected pull, memcpy or rccl", forced);
  } else {
    if (peer_ok) order.push_back(XCHG_PULL);
    order.push_back(XCHG_MEMCPY);
    if (distinct) order.push_back(XCHG_RCCL);
  }
  int last_rc = fail((int)ICICLE_UNKNOWN_ERROR, "no exchange transport available");
  std::string why;
  for (int m : order) {
    int rc = 0;
    if (m == XCHG_PULL && !peer_ok) rc = fail((int)ICICLE_INVALID_DEVICE, "pull exchange: peer access between the devices of the group is not available");
    if (m == XCHG_RCCL) {
      if (!distinct) rc = fail((int)ICICLE_INVALID_DEVICE, "rccl exchange: the devices of the group must be distinct");
      else if (!load_rccl(g->rccl)) rc = (int)ICICLE_BACKEND_LOAD_FAILED;
      else if (g->comms.empty()) {
        g->comms.assign(G, nullptr);
        if (g->rccl.init_all(G, g->devs.data(), g->comms.data())) {
          g->comms.clear();
          rc = fail((int)ICICLE_UNKNOWN_ERROR, "rccl exchange: %s", g->rccl.last_error());
        }
      }
    }
    if (!rc) {
      g->mode = m;
      rc = exchange_self_test(g);
    }
    if (!rc) {
      if (getenv("ICICLE_SNARK_VERBOSE")) fprintf(stderr, "[icicle-snark-hip] device group of %d: %s exchange\n", G, mode_name(m));
      return 0;
    }
    last_rc = rc;
    why += std::string(why.empty() ? "" : "; ") + mode_name(m) + ": " + last_error_text();
  }
  return fail(last_rc, "no working exchange for the device group (%s)", why.c_str());
}

// ------------------------------------------------------------------------------------------------ group build
int group_load(Groth16CacheManager* cm, const char* key, const uint8_t* zkey, size_t len, const std::vector<int>& devs)
{
  const int G = (int)devs.size();
  if (G < 2 || G > (int)XCHG_MAX_PEERS) return fail(ERR_ARG, "a device group has 2 to %u shards", XCHG_MAX_PEERS);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail((int)ICICLE_INVALID_DEVICE, "no HIP device available (this library has no CPU fallback)");
  for (int d : devs)
    if (d < 0 || d >= ndev) return fail((int)ICICLE_INVALID_DEVICE, "device %d of the group does not exist (%d visible)", d, ndev);
  std::shared_ptr<DeviceGroup> g(new DeviceGroup());
  g->devs = devs;
  g->shards.resize(G);
  g->peer_ok = enable_peer_access(devs);
  // one builder thread per distinct device; the shards of one device are built one after the other (they share its null
  // stream, staging pool and memory)
  std::vector<int> distinct;
  for (int d : devs)
    if (std::find(distinct.begin(), distinct.end(), d) == distinct.end()) distinct.push_back(d);
  RunState st;
  const void* hint_base;
  size_t hint_len;
  int hint_fd;
  staged_copy_file_hint_get(&hint_base, &hint_len, &hint_fd);
  {
    std::vector<std::thread> builders;
    for (int d : distinct)
      builders.emplace_back([&, d] {
        staged_copy_file_hint(hint_base, hint_len, hint_fd);
        for (int r = 0; r < G; r++) {
          if (devs[r] != d || st.failed) continue;
          std::unique_ptr<ZKeyCache> z;
          const int rc = build_cache(zkey, len, d, r, G, z);
          st.note(rc);
          if (!rc) {
            z->in_group = true;
            g->shards[r] = std::shared_ptr<ZKeyCache>(z.release());
          }
        }
      });
    for (auto& t : builders) t.join();
  }
  if (int rc = st.finish()) return rc;
  g->dist = true;
  for (auto& s : g->shards) g->dist = g->dist && shard_dist_supported(s.get());
  g->ev_slice.assign(G, nullptr);
  g->ev_s1.assign(G, nullptr);
  g->ev_s2.assign(G, nullptr);
  for (int r = 0; r < G; r++) {
    P_HIP(hipSetDevice(devs[r]));
    for (auto* v : {&g->ev_slice, &g->ev_s1, &g->ev_s2}) P_HIP(hipEventCreateWithFlags(&(*v)[r], hipEventDisableTiming));
  }
  g->team.reset(new Team(G));
  if (int rc = setup_exchange(g.get())) return rc;
  (void)set_active_device(devs[0]);
  std::lock_guard<std::mutex> lm(cm->map_mu);
  cm->groups[key] = g;
  return 0;
}

// ------------------------------------------------------------------------------------------------ group prove
// caller holds cm->mu.  out_points = the SUM of the shards' commitments (what one device would have computed).
int group_commitments(Groth16CacheManager* cm, DeviceGroup* g, const void* wtns, size_t wtns_len, uint8_t* out_points, Groth16Timings* tm)
{
  const int G = (int)g->devs.size();
  const auto t0 = std::chrono::steady_clock::now();
  Wtns w;
  if (wtns) {
    if (int rc = parse_wtns((const uint8_t*)wtns, wtns_len, w)) return rc;
  } else
    for (auto& s : g->shards)
      if (!s->witness_resident) return fail(ERR_ARG, "no witness given and none resident on the devices");
  const void* hint_base;
  size_t hint_len;
  int hint_fd;
  staged_copy_file_hint_get(&hint_base, &hint_len, &hint_fd);
  std::vector<uint8_t> blocks((size_t)G * GROTH16_COMMITMENTS_BYTES);
  std::vector<Groth16Timings> tms(G);
  std::vector<double> up_ms(G, 0.0);
  std::vector<void*> wit(G), y(G), send2(G);
  for (int r = 0; r < G; r++) wit[r] = g->shards[r]->d_witness;
  RunState st;
  const bool dist = g->dist;
  g->team->run([&](int r) {
    ZKeyCache* z = g->shards[r].get();
    hipStream_t sq = z->s_qap;
    int rc = set_active_device(g->devs[r]);
    // 1. witness: 1/G over PCIe, the rest over the exchange
    if (wtns) {
      if (!rc && !st.failed) {
        const auto tu = std::chrono::steady_clock::now();
        staged_copy_file_hint(hint_base, hint_len, hint_fd);
        rc = shard_upload_slice(z, w);
        staged_copy_file_hint(nullptr, 0, -1);
        up_ms[r] = ms_since(tu);
        if (!rc && hipEventRecord(g->ev_slice[r], sq) != hipSuccess) rc = fail((int)ICICLE_UNKNOWN_ERROR, "hipEventRecord");
        // the shard's own slice is in place: its witness MSMs need no more than that (shard_commitments: own_slice_first)
        if (!rc && hipEventRecord(z->ev_own_slice, sq) != hipSuccess) rc = fail((int)ICICLE_UNKNOWN_ERROR, "hipEventRecord");
        z->own_slice_event_set = !rc;
      }
      st.note(rc);
      g->team->barrier(); // every ev_slice is recorded
      if (!st.failed) {
        rc = xchg_allgather(g, r, wit.data(), (size_t)witness_slice_elems(z->n_vars, G) * 32, g->ev_slice.data(), sq);
        if (!rc && hipEventRecord(z->ev_witness, sq) != hipSuccess) rc = fail((int)ICICLE_UNKNOWN_ERROR, "hipEventRecord");
        if (!rc) z->witness_resident = true;
        st.note(rc);
      }
    }
    // (the witness sort and MSM chains of shard_commitments wait for ev_witness: with a resident witness it is recorded there)
    z->witness_event_set = wtns != nullptr && !st.failed;
    // 2./3. distributed front end
    if (dist) {
      if (!st.failed) {
        rc = shard_dist_stage1(cm, z);
        if (!rc && hipEventRecord(g->ev_s1[r], sq) != hipSuccess) rc = fail((int)ICICLE_UNKNOWN_ERROR, "hipEventRecord");
        st.note(rc);
      }
      y[r] = z->d_dist_y;
      send2[r] = z->d_dist_send2;
      g->team->barrier();
      const uint32_t m = z->domain_size / (uint32_t)G;
      if (!st.failed) {
        rc = xchg_alltoall(g, r, y.data(), z->d_dist_recv1, 3, (size_t)m * 32, (size_t)(m / G) * 32, g->ev_s1.data(), sq);
        if (!rc) rc = shard_dist_stage2(z);
        if (!rc && hipEventRecord(g->ev_s2[r], sq) != hipSuccess) rc = fail((int)ICICLE_UNKNOWN_ERROR, "hipEventRecord");
        st.note(rc);
      }
      send2[r] = z->d_dist_send2;
      g->team->barrier();
      if (!st.failed) {
        rc = xchg_alltoall(g, r, send2.data(), z->d_fold, 3, (size_t)m * 32, (size_t)(m / G) * 32, g->ev_s2.data(), sq);
        if (!rc) z->dist_ready = true; // exchange 2 is enqueued on the stream the forward transform follows on
        st.note(rc);
      }
    }
    // 4. the rest of this shard's pipeline: forward transform, digit sorts, five MSMs, Horner tails on host threads
    if (!st.failed) {
      rc = shard_commitments(cm, z, nullptr, 0, blocks.data() + (size_t)r * GROTH16_COMMITMENTS_BYTES, &tms[r], nullptr);
      st.note(rc);
    }
    if (st.failed) {
      // leave nothing in flight that a peer's buffers (or this shard's) are part of
      for (hipStream_t s : {z->s_qap, z->s_g1, z->s_g2, z->s_g3, z->s_g4, z->s_g5})
        if (s) (void)hipStreamSynchronize(s);
      z->dist_ready = z->dist_stage2_done = false;
      z->witness_event_set = false;
    }
    z->own_slice_event_set = false;
    g->team->barrier(); // no shard returns (and lets the next prove overwrite its buffers) while a peer may still read them
  });
  (void)set_active_device(g->devs[0]);
  if (int rc = st.finish()) return rc;
  if (int rc = groth16_sum_commitments(blocks.data(), G, out_points)) return rc;
  // phase times of the slowest shard; the upload is a host-side time (the all-gather belongs to the QAP phase's stream)
  Groth16Timings agg = {0, 0, 0, 0};
  for (int r = 0; r < G; r++) {
    agg.h2d_ms = std::max(agg.h2d_ms, up_ms[r]);
    agg.qap_ms = std::max(agg.qap_ms, tms[r].qap_ms);
    agg.msm_ms = std::max(agg.msm_ms, tms[r].msm_ms);
  }
  agg.total_ms = ms_since(t0);
  g->shards[0]->last_tm = agg;
  if (tm) *tm = agg;
  return 0;
}

} // namespace prover
} // namespace isnark
