// rccl_loopback.cpp -- TEST INFRASTRUCTURE: the nccl* entry points libcssm_pf resolves at run time (cssm_shard.hip: rccl_api), for ranks
// that are THREADS of one process on one GPU.  RCCL admits one rank per device and process, so on the one-GPU test box the library's own
// series loop (cssm_pf_shard_series_rccl: kernels and collectives enqueued by the library, one stream per rank) could only ever run at
// world 1; with CSSM_RCCL_LIB pointing here it runs at world 2 .. 8 -- the same buffers, counts, displacements and call order a node would
// see, the data moved by the device-to-device copies a collective amounts to:
//   ncclAllToAll   recv[q * count ..] <- rank q's send[rank * count ..]
//   ncclAllToAllv  recv[rdispls[q] ..] <- rank q's send[sdispls_q[rank] ..], sendcounts_q[rank] elements (checked against recvcounts[q])
//   ncclAllGather  recv[q * count ..] <- rank q's send
// Ordering: every rank records an event behind what produced its send buffer, the ranks meet (a host barrier of the threads), every rank
// makes ITS stream wait for the peers' events and enqueues its copies, records a second event, the ranks meet again and every stream waits
// for the peers' second events -- no rank overwrites a send buffer a peer has not read.  What this cannot show is a link: the
// transport.  Built by tests/test_gpu_rccl_loopback.py:  hipcc -O2 -shared -fPIC --offload-arch=gfx950 -o librccl_loopback.so rccl_loopback.cpp
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
struct Group {
  int world = 0, joined = 0, left = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long gen = 0;
  bool aborted = false;
  const void* send[64];
  const size_t* scounts[64];
  const size_t* sdispls[64];
  hipEvent_t ready[64], done[64];
};
struct Comm { Group* g; int rank; };
std::mutex g_reg;
std::map<std::string, Group*> g_groups;
unsigned long long g_next_id = 1;

bool meet(Group* g) {   // false: the communicator was aborted
  std::unique_lock<std::mutex> lk(g->m);
  if (g->aborted) return false;
  const long my = g->gen;
  if (++g->arrived == g->world) { g->arrived = 0; ++g->gen; g->cv.notify_all(); return true; }
  g->cv.wait(lk, [&] { return g->gen != my || g->aborted; });
  return !g->aborted;
}
size_t elem(int dtype) { return (dtype == 5 || dtype == 8 || dtype == 4) ? 8 : ((dtype == 2 || dtype == 3 || dtype == 7) ? 4 : 1); }   // ncclDataType_t (rccl.h)

template <class Copy>
int collective(Comm* c, const void* send, const size_t* scounts, const size_t* sdispls, hipStream_t stream, Copy copy) {
  Group* g = c->g;
  g->send[c->rank] = send; g->scounts[c->rank] = scounts; g->sdispls[c->rank] = sdispls;
  if (hipEventRecord(g->ready[c->rank], stream) != hipSuccess) return 1;
  if (!meet(g)) return 6;
  for (int q = 0; q < g->world; ++q) {
    if (hipStreamWaitEvent(stream, g->ready[q], 0) != hipSuccess) return 1;
    if (copy(q) != hipSuccess) return 1;
  }
  if (hipEventRecord(g->done[c->rank], stream) != hipSuccess) return 1;
  if (!meet(g)) return 6;
  for (int q = 0; q < g->world; ++q)
    if (hipStreamWaitEvent(stream, g->done[q], 0) != hipSuccess) return 1;
  return meet(g) ? 0 : 6;   // (the published pointers stay valid until every rank has read them)
}
}  // namespace

extern "C" {
int ncclGetUniqueId(void* id) {
  std::lock_guard<std::mutex> lk(g_reg);
  memset(id, 0, 128);
  const unsigned long long v = g_next_id++;
  memcpy(id, "LOOPBACK", 8);
  memcpy((char*)id + 8, &v, 8);
  return 0;
}
struct Id128 { char b[128]; };
int ncclCommInitRank(void** comm, int world, Id128 id, int rank) {
  if (world < 1 || world > 64 || rank < 0 || rank >= world || memcmp(id.b, "LOOPBACK", 8) != 0) return 4;
  Group* g = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_reg);
    const std::string key(id.b, 16);
    auto it = g_groups.find(key);
    if (it == g_groups.end()) { g = new Group(); g->world = world; g_groups[key] = g; } else g = it->second;
    if (g->world != world) return 4;
    ++g->joined;
  }
  if (hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming) != hipSuccess) return 1;
  Comm* c = new Comm{g, rank};
  if (!meet(g)) return 6;       // (as ncclCommInitRank: returns when every rank has joined)
  *comm = c;
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  (void)hipEventDestroy(c->g->ready[c->rank]); (void)hipEventDestroy(c->g->done[c->rank]);
  delete c;                     // (the group object stays: a few hundred bytes per test communicator)
  return 0;
}
int ncclCommAbort(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  { std::lock_guard<std::mutex> lk(c->g->m); c->g->aborted = true; }
  c->g->cv.notify_all();
  return 0;
}
int ncclCommGetAsyncError(void* comm, int* err) { *err = (comm && static_cast<Comm*>(comm)->g->aborted) ? 6 : 0; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 6 ? "loopback communicator aborted" : (r == 4 ? "invalid argument (loopback)" : "HIP error inside the loopback collective")); }

int ncclAllToAll(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t e = elem(dtype), r = (size_t)c->rank;
  return collective(c, send, nullptr, nullptr, stream, [&](int q) {
    return hipMemcpyAsync((char*)recv + (size_t)q * count * e, (const char*)c->g->send[q] + r * count * e, count * e, hipMemcpyDeviceToDevice, stream);
  });
}
int ncclAllToAllv(const void* send, const size_t* scounts, const size_t* sdispls, void* recv, const size_t* rcounts, const size_t* rdispls, int dtype,
                  void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t e = elem(dtype), r = (size_t)c->rank;
  int mismatch = 0;
  const int rc = collective(c, send, scounts, sdispls, stream, [&](int q) {
    const size_t n = c->g->scounts[q][r];
    if (n != rcounts[q]) { mismatch = 1; return hipSuccess; }   // (what RCCL would turn into a hang or a fault: reported instead)
    if (n == 0) return hipSuccess;
    return hipMemcpyAsync((char*)recv + rdispls[q] * e, (const char*)c->g->send[q] + c->g->sdispls[q][r] * e, n * e, hipMemcpyDeviceToDevice, stream);
  });
  return rc ? rc : (mismatch ? 4 : 0);
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t e = elem(dtype);
  return collective(c, send, nullptr, nullptr, stream, [&](int q) {
    return hipMemcpyAsync((char*)recv + (size_t)q * count * e, c->g->send[q], count * e, hipMemcpyDeviceToDevice, stream);
  });
}
}
