// stub_rccl.cpp -- a stand-in librccl for TESTS (never shipped, never linked by libmhx.so).
//
// libmhx binds eight nccl* entry points by dlopen (csrc/mhx_engine.cpp, struct Rccl).  On the one
// GPU a test box has, the real library cannot run a communicator of more than one rank inside one
// process (it wants a device per rank), so the branch a multi-GPU host takes - ncclCommInitAll,
// one ncclAllReduce per engine between ncclGroupStart and ncclGroupEnd, each on its engine's own
// stream - would never execute before the first 8-GPU run.  This library implements those eight
// symbols with the NCCL calling conventions (opaque communicator, 128-byte id, result codes,
// grouped calls take effect at ncclGroupEnd, the collective is ordered on the caller's stream) and
// a host-staged sum over the communicators of ONE process, in rank order.  Selected with
// MHX_RCCL_LIBRARY=<this .so>; MHX_GROUP_FORCE_RCCL=1 lets a group whose engines share a device
// take the communicator branch.
//
//   MHX_STUB_RCCL_LOG=<file>   one line per call: what libmhx asked for, with the device that was
//                              current at the time (tests assert the call pattern from it)
//   MHX_STUB_RCCL_FAIL=allreduce|groupend|initall   the named call returns ncclInternalError
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
enum { kSuccess = 0, kUnhandledDeviceError = 1, kSystemError = 2, kInternalError = 3,
       kInvalidArgument = 4, kInvalidUsage = 5 };
constexpr int kDouble = 8, kSum = 0;  // ncclDouble (= ncclFloat64), ncclSum

struct Clique {
  int n = 0;
  std::vector<struct Comm*> members;
};
struct Comm {
  unsigned magic = 0x5CC1;
  Clique* clique = nullptr;
  int rank = 0, device = 0;
};
struct Pending {
  Comm* comm;
  const void* send;
  void* recv;
  size_t count;
  hipStream_t stream;
};

std::mutex g_mu;
int g_depth = 0;
std::vector<Pending> g_pending;
std::map<unsigned long long, Clique*> g_by_id;
unsigned long long g_next_id = 1;

bool fail_at(const char* what) {
  const char* f = getenv("MHX_STUB_RCCL_FAIL");
  return f && strcmp(f, what) == 0;
}
void logf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void logf(const char* fmt, ...) {
  const char* path = getenv("MHX_STUB_RCCL_LOG");
  if (!path || !*path) return;
  FILE* f = fopen(path, "a");
  if (!f) return;
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fputc('\n', f);
  fclose(f);
}
int current_device() {
  int d = -1;
  (void)hipGetDevice(&d);
  return d;
}

// every rank of a clique has posted `count` doubles: sum in rank order, hand the sum to all.
// Stream-ordered the simple way: wait for each rank's stream (what was enqueued before the
// collective has run), stage through the host, copy back before returning.
int run_allreduce(Clique* q, const std::vector<Pending>& ops) {
  const size_t count = ops[0].count;
  int keep = current_device();
  std::vector<double> sum(count, 0.0), h(count);
  for (int r = 0; r < q->n; ++r) {
    const Pending* p = nullptr;
    for (const Pending& o : ops)
      if (o.comm->rank == r) p = &o;
    if (!p || p->count != count) return kInvalidUsage;
    if (hipSetDevice(p->comm->device) != hipSuccess) return kUnhandledDeviceError;
    if (hipStreamSynchronize(p->stream) != hipSuccess) return kUnhandledDeviceError;
    if (hipMemcpy(h.data(), p->send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      return kUnhandledDeviceError;
    for (size_t i = 0; i < count; ++i) sum[i] += h[i];
  }
  for (const Pending& o : ops) {
    if (hipSetDevice(o.comm->device) != hipSuccess) return kUnhandledDeviceError;
    if (hipMemcpy(o.recv, sum.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      return kUnhandledDeviceError;
  }
  if (keep >= 0) (void)hipSetDevice(keep);  // (NCCL leaves the caller's device as it found it)
  return kSuccess;
}

int flush() {
  std::vector<Pending> ops;
  ops.swap(g_pending);
  std::map<Clique*, std::vector<Pending>> by;
  for (const Pending& p : ops) by[p.comm->clique].push_back(p);
  for (auto& kv : by) {
    if ((int)kv.second.size() != kv.first->n) {
      logf("flush clique_of=%d posted=%zu INCOMPLETE", kv.first->n, kv.second.size());
      return kInvalidUsage;  // a rank of this process never posted: the real library would hang
    }
    const int rc = run_allreduce(kv.first, kv.second);
    logf("flush clique_of=%d count=%zu rc=%d", kv.first->n, kv.second[0].count, rc);
    if (rc != kSuccess) return rc;
  }
  return kSuccess;
}
}  // namespace

extern "C" {

struct ncclUniqueIdStub { char internal[128]; };

int ncclGetUniqueId(ncclUniqueIdStub* id) {
  if (!id) return kInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id->internal, 0, sizeof id->internal);
  const unsigned long long v = g_next_id++;
  memcpy(id->internal, "MHXSTUB", 8);
  memcpy(id->internal + 8, &v, sizeof v);
  logf("GetUniqueId id=%llu", v);
  return kSuccess;
}

int ncclCommInitRank(Comm** comm, int nranks, ncclUniqueIdStub id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
  if (memcmp(id.internal, "MHXSTUB", 8) != 0) return kInvalidArgument;
  unsigned long long v = 0;
  memcpy(&v, id.internal + 8, sizeof v);
  std::lock_guard<std::mutex> lk(g_mu);
  Clique*& q = g_by_id[v];
  if (!q) {
    q = new Clique();
    q->n = nranks;
    q->members.assign((size_t)nranks, nullptr);
  }
  if (q->n != nranks || q->members[(size_t)rank]) return kInvalidUsage;
  Comm* c = new Comm();
  c->clique = q;
  c->rank = rank;
  c->device = current_device();
  q->members[(size_t)rank] = c;
  *comm = c;
  logf("CommInitRank id=%llu nranks=%d rank=%d device=%d", v, nranks, rank, c->device);
  return kSuccess;
}

int ncclCommInitAll(Comm** comms, int ndev, const int* devlist) {
  if (!comms || ndev < 1) return kInvalidArgument;
  if (fail_at("initall")) {
    logf("CommInitAll ndev=%d FAIL(injected)", ndev);
    return kInternalError;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  Clique* q = new Clique();
  q->n = ndev;
  char devs[256] = "";
  for (int i = 0; i < ndev; ++i) {
    Comm* c = new Comm();
    c->clique = q;
    c->rank = i;
    c->device = devlist ? devlist[i] : i;
    q->members.push_back(c);
    comms[i] = c;
    const size_t at = strlen(devs);
    snprintf(devs + at, sizeof devs - at, "%s%d", i ? "," : "", c->device);
  }
  logf("CommInitAll ndev=%d devices=%s", ndev, devs);
  return kSuccess;
}

int ncclCommDestroy(Comm* c) {
  if (!c || c->magic != 0x5CC1) return kInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mu);
  logf("CommDestroy rank=%d device=%d", c->rank, c->device);
  if (c->clique) {
    c->clique->members[(size_t)c->rank] = nullptr;
    bool empty = true;
    for (Comm* m : c->clique->members) empty = empty && !m;
    if (empty) {
      for (auto it = g_by_id.begin(); it != g_by_id.end();)
        it = it->second == c->clique ? g_by_id.erase(it) : ++it;
      delete c->clique;
    }
  }
  c->magic = 0;
  delete c;
  return kSuccess;
}

int ncclGroupStart() {
  std::lock_guard<std::mutex> lk(g_mu);
  ++g_depth;
  logf("GroupStart depth=%d", g_depth);
  return kSuccess;
}

int ncclGroupEnd() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_depth <= 0) return kInvalidUsage;
  --g_depth;
  logf("GroupEnd depth=%d pending=%zu", g_depth, g_pending.size());
  if (g_depth > 0) return kSuccess;
  if (fail_at("groupend")) {
    g_pending.clear();
    return kInternalError;
  }
  return flush();
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, Comm* c,
                  hipStream_t stream) {
  if (!c || c->magic != 0x5CC1 || !send || !recv) return kInvalidArgument;
  if (dtype != kDouble || op != kSum) return kInvalidArgument;  // all libmhx ever asks for
  std::lock_guard<std::mutex> lk(g_mu);
  logf("AllReduce rank=%d comm_device=%d current_device=%d count=%zu in_group=%d", c->rank,
       c->device, current_device(), count, g_depth > 0 ? 1 : 0);
  if (fail_at("allreduce")) return kInternalError;
  g_pending.push_back(Pending{c, send, recv, count, stream});
  if (g_depth > 0) return kSuccess;
  return flush();  // ungrouped: complete only for a clique of one (else INCOMPLETE -> error)
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case kSuccess: return "no error";
    case kUnhandledDeviceError: return "unhandled device error (stub)";
    case kSystemError: return "unhandled system error (stub)";
    case kInternalError: return "internal error (stub)";
    case kInvalidArgument: return "invalid argument (stub)";
    case kInvalidUsage: return "invalid usage (stub)";
    default: return "unknown result code (stub)";
  }
}

}  // extern "C"
