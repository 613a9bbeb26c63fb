// rccl_stub.cpp — TEST INFRASTRUCTURE: a stand-in for librccl.so that lets SEVERAL RANKS SHARE ONE GPU.
//
// RCCL refuses a communicator with two ranks on one device ("Duplicate GPU detected"), and the builder's boxes have one GPU: the
// N > 1 code of the C-ABI (to_comm_init_rank with nranks > 1, the in-place ncclAllGather of equal shards, the grouped ncclBroadcast of
// unequal ones, to_allgather_stats; csrc/trajopt_hip.hip) would otherwise never execute.  libtrajopt_hip.so dlopen()s its collective
// library by name (TRAJOPT_RCCL_LIB overrides it); this file implements the seven entry points it binds — with the call signatures of
// rccl.h — over a POSIX shared-memory segment: every collective stages device -> segment -> device with a process barrier on either
// side.  No xGMI, no performance claim: the point is that the library's own rank / offset / count arithmetic and call sequence run with
// more than one rank and are checked bit for bit against a one-handle solve (tests/test_gpu_multi.py).
//
//   g++ -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rccl_stub.cpp -o librccl_stub.so -L/opt/rocm/lib -lamdhip64 -lrt
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr size_t kData = size_t(256) << 20;  // staging area (the tests move a few MB)
constexpr int kMaxRanks = 16;
constexpr double kTimeoutS = 120.0;

struct Segment {
  std::atomic<int> arrived;     // ranks that have attached
  std::atomic<int> bar_count;   // sense-reversing barrier
  std::atomic<int> bar_sense;
  std::atomic<int> detached;
  int nranks;
  char pad[64 - 5 * sizeof(int)];
  unsigned char data[kData];
};

struct Comm {
  Segment* seg = nullptr;
  int nranks = 0, rank = 0, sense = 0;
  char name[128];
};

struct UniqueId { char b[128]; };

enum { kSuccess = 0, kSystemError = 2, kInvalidArgument = 4, kInvalidUsage = 5 };

size_t dtype_size(int dt) {
  switch (dt) {
    case 0: case 1: return 1;            // int8, uint8
    case 2: case 3: case 7: return 4;    // int32, uint32, float32
    case 4: case 5: case 8: return 8;    // int64, uint64, float64
    case 6: case 9: return 2;            // float16, bfloat16
    default: return 0;
  }
}

bool barrier(Comm* c) {
  Segment* s = c->seg;
  const int want = c->sense ^ 1;
  if (s->bar_count.fetch_add(1) + 1 == c->nranks) {
    s->bar_count.store(0);
    s->bar_sense.store(want);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    while (s->bar_sense.load() != want) {
      std::this_thread::yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
    }
  }
  c->sense = want;
  return true;
}

// grouped calls (ncclGroupStart .. ncclGroupEnd): recorded, executed in order at the end — every rank records the same sequence
struct Op { int kind; const void* send; void* recv; size_t count; int dt; int root; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

int do_allgather(const Op& o) {
  Comm* c = o.comm;
  const size_t bytes = o.count * dtype_size(o.dt);
  if (bytes == 0 || bytes * c->nranks > kData) return kInvalidArgument;
  if (hipStreamSynchronize(o.stream) != hipSuccess) return kSystemError;
  if (hipMemcpy(c->seg->data + bytes * c->rank, o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
  if (!barrier(c)) return kSystemError;
  // (in place or not: every block is written, the own one included — it is the same bytes)
  if (hipMemcpy(o.recv, c->seg->data, bytes * c->nranks, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
  return barrier(c) ? kSuccess : kSystemError;
}

int do_broadcast(const Op& o) {
  Comm* c = o.comm;
  const size_t bytes = o.count * dtype_size(o.dt);
  if (bytes == 0 || bytes > kData || o.root < 0 || o.root >= c->nranks) return kInvalidArgument;
  if (hipStreamSynchronize(o.stream) != hipSuccess) return kSystemError;
  if (c->rank == o.root && hipMemcpy(c->seg->data, o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
  if (!barrier(c)) return kSystemError;
  if (c->rank != o.root) {
    if (hipMemcpy(o.recv, c->seg->data, bytes, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
  } else if (o.recv != o.send) {
    if (hipMemcpy(o.recv, o.send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return kSystemError;
  }
  return barrier(c) ? kSuccess : kSystemError;
}

int run(const Op& o) { return o.kind == 0 ? do_allgather(o) : do_broadcast(o); }

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  if (!id) return kInvalidArgument;
  std::memset(id->b, 0, sizeof(id->b));
  std::snprintf(id->b, sizeof(id->b), "/trajopt_rccl_stub_%d_%lld", (int)getpid(),
                (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return kSuccess;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.b[0] != '/') return kInvalidArgument;
  int fd = -1;
  const auto t0 = std::chrono::steady_clock::now();
  bool creator = false;
  while (fd < 0) {  // whoever comes first creates the segment; the others find it
    fd = shm_open(id.b, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd >= 0) { creator = true; break; }
    if (errno != EEXIST) return kSystemError;
    fd = shm_open(id.b, O_RDWR, 0600);
    if (fd < 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return kSystemError;
  }
  if (creator && ftruncate(fd, sizeof(Segment)) != 0) { close(fd); shm_unlink(id.b); return kSystemError; }
  if (!creator) {  // wait until the creator has sized it
    off_t sz = 0;
    while ((sz = lseek(fd, 0, SEEK_END)) < (off_t)sizeof(Segment)) {
      std::this_thread::yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) { close(fd); return kSystemError; }
    }
  }
  void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return kSystemError;
  Comm* c = new Comm();
  c->seg = static_cast<Segment*>(p);   // a fresh segment is zero-filled: counters start at 0, sense 0
  c->nranks = nranks; c->rank = rank; c->sense = 0;
  std::snprintf(c->name, sizeof(c->name), "%s", id.b);
  c->seg->arrived.fetch_add(1);
  while (c->seg->arrived.load() < nranks) {
    std::this_thread::yield();
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) { munmap(p, sizeof(Segment)); delete c; return kSystemError; }
  }
  *comm = c;
  return kSuccess;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dt, void* comm, hipStream_t stream) {
  if (!comm || !send || !recv || dtype_size(dt) == 0) return kInvalidArgument;
  Op o{0, send, recv, count, dt, 0, static_cast<Comm*>(comm), stream};
  if (g_depth > 0) { g_ops.push_back(o); return kSuccess; }
  return run(o);
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dt, int root, void* comm, hipStream_t stream) {
  if (!comm || !send || !recv || dtype_size(dt) == 0) return kInvalidArgument;
  Op o{1, send, recv, count, dt, root, static_cast<Comm*>(comm), stream};
  if (g_depth > 0) { g_ops.push_back(o); return kSuccess; }
  return run(o);
}

int ncclGroupStart(void) { ++g_depth; return kSuccess; }

int ncclGroupEnd(void) {
  if (g_depth <= 0) return kInvalidUsage;
  if (--g_depth > 0) return kSuccess;
  int rc = kSuccess;
  for (const Op& o : g_ops) { rc = run(o); if (rc != kSuccess) break; }
  g_ops.clear();
  return rc;
}

int ncclCommDestroy(void* comm) {
  if (!comm) return kInvalidArgument;
  Comm* c = static_cast<Comm*>(comm);
  if (c->seg->detached.fetch_add(1) + 1 == c->nranks) shm_unlink(c->name);  // the last one out removes the name
  munmap(c->seg, sizeof(Segment));
  delete c;
  return kSuccess;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case kSuccess: return "no error";
    case kSystemError: return "rccl stub: system error (shared memory, HIP copy, or a peer that never arrived)";
    case kInvalidArgument: return "rccl stub: invalid argument";
    case kInvalidUsage: return "rccl stub: invalid usage";
    default: return "rccl stub: unknown error";
  }
}

}  // extern "C"
