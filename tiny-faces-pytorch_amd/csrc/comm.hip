// comm: the gradient exchange of the data-parallel path over RCCL, behind the C ABI (tf_comm_* / tf_allreduce_bucket).
//
// The reference has no distributed code; the 8-GPU path is one process per GPU, gradients SUMMED over the ranks in a few
// large buckets while the backward pass is still running, the average folded into the SGD step (tf_sgd_step grad_scale).  Rounds 1-3 issued the
// collectives through torch.distributed from a ctypes callback inside the backward enqueue: a maintainer binding only this library had no
// data-parallel path, and the enqueue took the GIL nine times per step.  Here:
//   * librccl.so is resolved at RUN time (dlopen / dlsym; first the copy already mapped into the process -- PyTorch ships one -- so there is a
//     single RCCL instance): the library has no link-time dependency on it and loads on hosts without RCCL;
//   * tf_comm = { ncclComm_t, a communication stream of its own at the DEFAULT priority (a low-priority stream next to RCCL halved the step
//     in round 3: profiles/r03_stream_priority.txt), a pool of events };
//   * tf_allreduce_bucket(comm, buf, n, after): in-place SUM of n floats on the communication stream, ordered behind the current tail of
//     `after` -- the executor's stream that carries the bucket's last gradient kernel -- without holding that stream up;
//   * tf_comm_allreduce_hook: a ready-made tf_grad_ready_fn (tf_detnet_hooks.fn) that reduces the slice of the flat gradient a bucket owns
//     (tf_comm_plan: block -> [start, end) elements); tf_comm_join makes the training stream wait for every collective issued so far.
// xGMI is a point-to-point mesh (7 links per GPU): each bucket is ONE large message so that RCCL can spread it over all links; nothing is copied.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "tuning.h"

namespace {

typedef int ncclResult_t;
typedef void* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
constexpr int kNcclFloat32 = 7, kNcclSum = 0;       // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.lib) break; }      // the copy the process already mapped
    for (const char* n : names) { if (r.lib) break; r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!r.lib) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
  });
  return r;
}

}  // namespace

struct tf_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, world = 1, device = 0;
  std::vector<hipEvent_t> events; size_t ev_next = 0;
  hipEvent_t next_event() {
    if (ev_next == events.size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; events.push_back(e); }
    return events[ev_next++];
  }
};

extern "C" int tf_comm_available(void) { return rccl().ok ? 1 : 0; }

// rank 0 draws the identifier of a new communicator (128 bytes, HOST memory) and ships it to the other ranks by any out-of-band means
// (the Python surface broadcasts it through the torch.distributed store it already has)
extern "C" int tf_comm_unique_id(void* id_out) {
  if (!id_out) return TF_ERR_ARG;
  if (!rccl().ok) return TF_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != 0) return TF_ERR_LAUNCH;
  memcpy(id_out, id.internal, sizeof(id.internal));
  return TF_OK;
}

// collective over all `world` ranks; binds to the CURRENT device (one process per GPU)
extern "C" int tf_comm_init(const void* id128, int rank, int world, tf_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return TF_ERR_ARG;
  if (!rccl().ok) return TF_ERR_UNSUPPORTED;
  if (tf::tuning().comm_fail_init) return TF_ERR_LAUNCH;      // test knob: the fall-back of the engine to the torch.distributed exchange
  tf_comm* c = new tf_comm;
  c->rank = rank; c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return TF_ERR_LAUNCH; }
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  const ncclResult_t rc = rccl().CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) {
    fprintf(stderr, "tinyfaces: ncclCommInitRank failed: %s\n", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    (void)hipStreamDestroy(c->stream); delete c; return TF_ERR_LAUNCH;
  }
  *out = c;
  return TF_OK;
}

extern "C" int tf_comm_destroy(tf_comm* c) {
  if (!c) return TF_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TF_OK;
}

extern "C" int tf_comm_rank(const tf_comm* c) { return c ? c->rank : -1; }
extern "C" int tf_comm_world(const tf_comm* c) { return c ? c->world : 0; }

// buf[0 .. n) <- sum over ranks, in place, on the communicator's own stream behind the current tail of `after` (NULL: behind nothing)
extern "C" int tf_allreduce_bucket(tf_comm* c, float* buf, size_t n, void* after) {
  if (!c || !buf) return TF_ERR_ARG;
  if (n == 0) return TF_OK;
  if (after) {
    hipEvent_t e = c->next_event();
    if (!e || hipEventRecord(e, (hipStream_t)after) != hipSuccess || hipStreamWaitEvent(c->stream, e, 0) != hipSuccess) return TF_ERR_LAUNCH;
  }
  const ncclResult_t rc = rccl().AllReduce(buf, buf, n, kNcclFloat32, kNcclSum, c->comm, c->stream);
  if (rc != 0) {
    fprintf(stderr, "tinyfaces: ncclAllReduce failed: %s\n", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    return TF_ERR_LAUNCH;
  }
  return TF_OK;
}

// `stream` waits for every collective issued on the communicator so far (the training stream calls this before the SGD step); also the
// point where the event pool of a step is recycled
extern "C" int tf_comm_join(tf_comm* c, void* stream) {
  if (!c) return TF_ERR_ARG;
  hipEvent_t e = c->next_event();
  if (!e || hipEventRecord(e, c->stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, e, 0) != hipSuccess) return TF_ERR_LAUNCH;
  c->ev_next = 0;
  return TF_OK;
}

// the ready-made gradient hook: tf_detnet_hooks.fn = tf_comm_allreduce_hook, .user = a tf_comm_plan.  Called by the executor when the bucket
// of `block` is final at the tail of `stream`; reduces grad_flat[start[k], end[k]) of the entry with blocks[k] == block.
extern "C" void tf_comm_allreduce_hook(int block, void* stream, void* user) {
  tf_comm_plan* p = (tf_comm_plan*)user;
  if (!p || !p->comm || !p->grad_flat) return;
  const int fail_bucket = tf::tuning().comm_fail_bucket;      // test knob (same on every rank)
  for (int k = 0; k < p->n; ++k)
    if (p->blocks[k] == block) {
      const int rc = k == fail_bucket ? TF_ERR_LAUNCH
                                      : tf_allreduce_bucket((tf_comm*)p->comm, p->grad_flat + p->start[k], (size_t)(p->end[k] - p->start[k]), stream);
      if (rc != TF_OK && p->rc == TF_OK) p->rc = rc;
      if (p->status) p->status[k] = rc == TF_OK ? 1 : rc;      // r6: per bucket, so that the caller can reduce exactly the buckets that were not
      ++p->issued;
    }
}
