// TEST INFRASTRUCTURE -- a recording stand-in for librccl, loaded through glaze_amd/csrc/rccl_dl.h when
// GLAZE_RCCL_LIBRARY names it.  It lets the n >= 2 exchange of Renderer::reduce_peers (communicators from ncclCommInitAll, every
// call of one exchange inside one ncclGroup issued from one thread) run on a box with ONE GPU: all "ranks" live on that device
// (GLAZE_MULTI_LOOPBACK=rccl).  Semantics kept from the real library, so that a renderer that gets them wrong produces a wrong
// image or an error here too:
//   * a collective / send / recv only takes effect at the outermost ncclGroupEnd, and only if it is complete there (every rank of
//     the communicator clique posted its ncclReduce with the same count and root; every ncclSend has its ncclRecv);
//   * a multi-rank call outside a group from the clique's single thread would block for ever: reported as ncclInvalidUsage;
//   * the work is stream-ordered and asynchronous: data is read on the sender's stream position and written on the receiver's,
//     nothing is synchronised with the host -- whoever reads the result must wait for the ROOT's stream, and a sender may only
//     touch its buffer again after waiting for its own stream.
// Every entry point appends one JSON line to the file named by GLAZE_FAKE_RCCL_LOG; GLAZE_FAKE_RCCL_HANG=<function>:<k> makes the k-th call of an entry point hang for ever; GLAZE_FAKE_RCCL_FAIL=<function>:<k> makes
// the k-th call (1-based) of that function fail with ncclInternalError.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Comm {
  int rank, nranks, device, clique;
  bool alive;
};
struct Op {
  int kind;   // 0 reduce, 1 send, 2 recv
  Comm* comm;
  const void* send;
  void* recv;
  size_t count;
  int peer_or_root;
  hipStream_t stream;
};
std::mutex g_m;
int g_depth = 0, g_cliques = 0;
std::vector<Op> g_ops;
std::map<std::string, int> g_calls;
unsigned long long g_seq = 0;

void logf(const char* fn, const Comm* c, const void* send, const void* recv, size_t count, int peer, hipStream_t st, int result) {
  const char* path = getenv("GLAZE_FAKE_RCCL_LOG");
  if (!path) return;
  FILE* f = fopen(path, "a");
  if (!f) return;
  int cur = -1;
  (void)hipGetDevice(&cur);
  fprintf(f, "{\"seq\": %llu, \"fn\": \"%s\", \"rank\": %d, \"nranks\": %d, \"clique\": %d, \"comm_device\": %d, \"current_device\": %d, \"send\": %llu, \"recv\": %llu, "
             "\"count\": %llu, \"peer\": %d, \"stream\": %llu, \"group_depth\": %d, \"result\": %d}\n",
          g_seq++, fn, c ? c->rank : -1, c ? c->nranks : 0, c ? c->clique : -1, c ? c->device : -1, cur, (unsigned long long)(uintptr_t)send,
          (unsigned long long)(uintptr_t)recv, (unsigned long long)count, peer, (unsigned long long)(uintptr_t)st, g_depth, result);
  fclose(f);
}
bool should_fail(const char* fn) {
  const int k = ++g_calls[fn];
  // GLAZE_FAKE_RCCL_HANG=<function>:<k>: the k-th call of that entry point never returns -- a collective whose peer never arrives, what a
  // first contact between ranks that goes wrong looks like from the caller's side (bench.py's watchdog is tested against it)
  if (const char* hang = getenv("GLAZE_FAKE_RCCL_HANG")) {
    const char* colon = strchr(hang, ':');
    if (colon && strlen(fn) == (size_t)(colon - hang) && !strncmp(hang, fn, colon - hang) && atoi(colon + 1) == k)
      for (;;) sleep(1);
  }
  const char* spec = getenv("GLAZE_FAKE_RCCL_FAIL");
  if (!spec) return false;
  const char* colon = strchr(spec, ':');
  if (!colon) return false;
  return strlen(fn) == (size_t)(colon - spec) && !strncmp(spec, fn, colon - spec) && atoi(colon + 1) == k;
}

__global__ void k_add(float* dst, const float* src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = dst[i] + src[i];
}

// `after` continues only when everything enqueued on `before` so far is done
bool order(hipStream_t before, hipStream_t after) {
  if (before == after) return true;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
  bool ok = hipEventRecord(e, before) == hipSuccess && hipStreamWaitEvent(after, e, 0) == hipSuccess;
  (void)hipEventDestroy(e);
  return ok;
}

ncclResult_t run_group() {
  std::vector<Op> ops;
  ops.swap(g_ops);
  std::vector<char> used(ops.size(), 0);
  // sends meet their receives, in posting order per (source, destination) pair
  for (size_t i = 0; i < ops.size(); ++i) {
    if (ops[i].kind != 1) continue;
    const Op& s = ops[i];
    size_t j = 0;
    for (; j < ops.size(); ++j)
      if (!used[j] && ops[j].kind == 2 && ops[j].comm->clique == s.comm->clique && ops[j].comm->rank == s.peer_or_root && ops[j].peer_or_root == s.comm->rank) break;
    if (j == ops.size() || ops[j].count != s.count) return ncclInvalidUsage;   // the real library would hang or corrupt
    used[i] = used[j] = 1;
    const Op& r = ops[j];
    if (hipSetDevice(r.comm->device) != hipSuccess) return ncclUnhandledCudaError;
    if (!order(s.stream, r.stream)) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(r.recv, s.send, s.count * sizeof(float), hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!order(r.stream, s.stream)) return ncclUnhandledCudaError;   // the send is complete on ITS stream once the data has left
  }
  for (size_t j = 0; j < ops.size(); ++j)
    if (ops[j].kind == 2 && !used[j]) return ncclInvalidUsage;   // a receive nobody sends to
  // reduces: one per rank of the clique, same count and root
  for (size_t i = 0; i < ops.size(); ++i) {
    if (ops[i].kind != 0 || used[i]) continue;
    const int clique = ops[i].comm->clique, n = ops[i].comm->nranks, root = ops[i].peer_or_root;
    std::vector<const Op*> by_rank((size_t)n, nullptr);
    for (size_t j = i; j < ops.size(); ++j)
      if (ops[j].kind == 0 && !used[j] && ops[j].comm->clique == clique) {
        if (by_rank[(size_t)ops[j].comm->rank] || ops[j].count != ops[i].count || ops[j].peer_or_root != root) return ncclInvalidUsage;
        by_rank[(size_t)ops[j].comm->rank] = &ops[j];
        used[j] = 1;
      }
    for (const Op* o : by_rank)
      if (!o) return ncclInvalidUsage;   // a rank is missing: the real collective never completes
    const Op& r = *by_rank[(size_t)root];
    if (!r.recv) return ncclInvalidArgument;
    if (hipSetDevice(r.comm->device) != hipSuccess) return ncclUnhandledCudaError;
    const size_t cnt = r.count;
    if (r.recv != r.send && hipMemcpyAsync(r.recv, r.send, cnt * sizeof(float), hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (int k = 0; k < n; ++k) {   // rank order: a fixed summation order, like a ring's
      if (k == root) continue;
      const Op& p = *by_rank[(size_t)k];
      if (!order(p.stream, r.stream)) return ncclUnhandledCudaError;
      hipLaunchKernelGGL(k_add, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, r.stream, static_cast<float*>(r.recv), static_cast<const float*>(p.send), cnt);
      if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    }
    for (int k = 0; k < n; ++k)
      if (k != root && !order(r.stream, by_rank[(size_t)k]->stream)) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

ncclResult_t post(const char* fn, int kind, Comm* c, const void* send, void* recv, size_t count, ncclDataType_t dt, int peer, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_m);
  ncclResult_t r = ncclSuccess;
  if (should_fail(fn)) r = ncclInternalError;
  else if (!c || !c->alive || dt != ncclFloat || peer < 0 || peer >= c->nranks) r = ncclInvalidArgument;
  else if (g_depth == 0 && c->nranks > 1) r = ncclInvalidUsage;   // one thread, several ranks, no group: the real call never returns
  logf(fn, c, send, recv, count, peer, st, (int)r);
  if (r != ncclSuccess) return r;
  g_ops.push_back(Op{kind, c, send, recv, count, peer, st});
  if (g_depth == 0) return run_group();
  return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetVersion(int* v) {
  if (v) *v = 99999;   // not a version any RCCL reports: a result produced with the stand-in is recognisable
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidUsage: return "invalid usage (fake_rccl: incomplete or ungrouped operation)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInternalError: return "internal error (fake_rccl: injected failure)";
    default: return "unhandled cuda error (fake_rccl)";
  }
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
  std::lock_guard<std::mutex> l(g_m);
  const bool fail = should_fail("ncclCommInitAll");
  logf("ncclCommInitAll", nullptr, nullptr, nullptr, (size_t)n, -1, nullptr, fail ? (int)ncclInternalError : 0);
  if (fail) return ncclInternalError;
  if (!comms || n < 1) return ncclInvalidArgument;
  const int clique = g_cliques++;
  for (int i = 0; i < n; ++i) comms[i] = reinterpret_cast<ncclComm_t>(new Comm{i, n, devs ? devs[i] : i, clique, true});
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  std::lock_guard<std::mutex> l(g_m);
  Comm* c = reinterpret_cast<Comm*>(comm);
  logf("ncclCommDestroy", c, nullptr, nullptr, 0, -1, nullptr, c && c->alive ? 0 : (int)ncclInvalidArgument);
  if (!c || !c->alive) return ncclInvalidArgument;
  c->alive = false;   // kept allocated: a use after destroy is reported instead of crashing
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
  std::lock_guard<std::mutex> l(g_m);
  const bool fail = should_fail("ncclGroupStart");
  logf("ncclGroupStart", nullptr, nullptr, nullptr, 0, -1, nullptr, fail ? (int)ncclInternalError : 0);
  if (fail) return ncclInternalError;
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  std::lock_guard<std::mutex> l(g_m);
  if (g_depth == 0) {
    logf("ncclGroupEnd", nullptr, nullptr, nullptr, 0, -1, nullptr, (int)ncclInvalidUsage);
    return ncclInvalidUsage;
  }
  ncclResult_t r = ncclSuccess;
  if (--g_depth == 0) r = should_fail("ncclGroupEnd") ? (g_ops.clear(), ncclInternalError) : run_group();
  logf("ncclGroupEnd", nullptr, nullptr, nullptr, 0, -1, nullptr, (int)r);
  return r;
}
ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t st) {
  if (op != ncclSum) return ncclInvalidArgument;
  return post("ncclReduce", 0, reinterpret_cast<Comm*>(comm), send, recv, count, dt, root, st);
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  return post("ncclSend", 1, reinterpret_cast<Comm*>(comm), send, nullptr, count, dt, peer, st);
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  return post("ncclRecv", 2, reinterpret_cast<Comm*>(comm), nullptr, recv, count, dt, peer, st);
}
}
