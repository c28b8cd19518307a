// comm.hip -- data-parallel gradient exchange behind the C-ABI: mxdet_comm_t = RCCL communicator + side stream + events.
//
// Role in the reference: MXNet's kvstore push/pull of every parameter's gradient (/root/reference/README.md:37);
// here one fp32 all-reduce per contiguous bucket of the flat gradient arena, one process per GPU, over RCCL / xGMI
// (SURVEY.md sections 8a10, 8b, 8e). The collective runs on the communicator's own stream: the caller's stream -- the
// data-gradient chain of backward -- never waits for it; whoever consumes a bucket (the optimizer stream) waits for
// that bucket's ticket.
//
// RCCL is resolved at run time with dlopen/dlsym: the library has no link-time dependency on it (single-GPU users
// never load it), and a process that already mapped an RCCL (torch bundles one under the same SONAME) gets that copy
// instead of a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

static_assert(sizeof(ncclUniqueId) == MXDET_COMM_ID_BYTES, "MXDET_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

namespace mxdet {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = {0};
};

// resolved once per process; immutable afterwards (not "mutable global state": a cache of dlsym results)
static Rccl g_rccl;
static std::once_flag g_rccl_once;

static void resolve_rccl() {
  Rccl& r = g_rccl;
  // the copy the process already mapped (torch's), else the system one
  r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!r.handle) r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!r.handle) {
    snprintf(r.why, sizeof(r.why), "librccl.so.1 not found: %s", dlerror());
    return;
  }
  struct { const char* name; void** slot; } syms[] = {
      {"ncclGetUniqueId", (void**)&r.GetUniqueId},   {"ncclCommInitRank", (void**)&r.CommInitRank},
      {"ncclCommDestroy", (void**)&r.CommDestroy},   {"ncclAllReduce", (void**)&r.AllReduce},
      {"ncclBroadcast", (void**)&r.Broadcast},       {"ncclGetErrorString", (void**)&r.GetErrorString},
  };
  for (auto& s : syms) {
    *s.slot = dlsym(r.handle, s.name);
    if (!*s.slot) {
      snprintf(r.why, sizeof(r.why), "librccl.so.1 lacks %s", s.name);
      return;
    }
  }
  r.ok = true;
}

static int need_rccl(const char* who) {
  std::call_once(g_rccl_once, resolve_rccl);
  MXDET_REQUIRE(g_rccl.ok, MXDET_ERCCL, "%s: RCCL unavailable (%s)", who, g_rccl.why);
  return MXDET_OK;
}

#define MXDET_RCCL(call, who)                                                              \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      ::mxdet::set_error("%s: RCCL: %s", who, ::mxdet::g_rccl.GetErrorString(r_));         \
      return MXDET_ERCCL;                                                                  \
    }                                                                                      \
  } while (0)

#define MXDET_HIP(call, who)                                            \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) {                                             \
      ::mxdet::set_error("%s: %s", who, hipGetErrorString(e_));         \
      return MXDET_EHIP;                                                \
    }                                                                   \
  } while (0)

}  // namespace mxdet

using namespace mxdet;

struct mxdet_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = -1;
  hipStream_t side = nullptr;                          // the collectives' own stream
  hipEvent_t ready[MXDET_COMM_MAX_INFLIGHT];           // "bucket final on the caller's stream"
  hipEvent_t done[MXDET_COMM_MAX_INFLIGHT];            // "bucket summed"
  long long issued = 0;                                // tickets handed out so far
};

extern "C" int mxdet_comm_unique_id(uint8_t* id) {
  clear_error();
  MXDET_REQUIRE(id != nullptr, MXDET_EINVAL, "comm_unique_id: null pointer");
  int rc = need_rccl("comm_unique_id");
  if (rc) return rc;
  ncclUniqueId u;
  MXDET_RCCL(g_rccl.GetUniqueId(&u), "comm_unique_id");
  memcpy(id, &u, MXDET_COMM_ID_BYTES);
  return MXDET_OK;
}

extern "C" int mxdet_comm_create(const uint8_t* id, int32_t world, int32_t rank, mxdet_comm_t** comm_out) {
  clear_error();
  MXDET_REQUIRE(id && comm_out, MXDET_EINVAL, "comm_create: null pointer");
  MXDET_REQUIRE(world >= 1 && rank >= 0 && rank < world, MXDET_EINVAL, "comm_create: rank %d of %d", rank, world);
  *comm_out = nullptr;
  int rc = need_rccl("comm_create");
  if (rc) return rc;
  mxdet_comm* c = new (std::nothrow) mxdet_comm();
  MXDET_REQUIRE(c != nullptr, MXDET_EINVAL, "comm_create: out of host memory");
  c->world = world; c->rank = rank;
  for (int i = 0; i < MXDET_COMM_MAX_INFLIGHT; ++i) c->ready[i] = c->done[i] = nullptr;
  auto fail = [&](int code) { mxdet_comm_destroy(c); return code; };
  if (hipGetDevice(&c->device) != hipSuccess) { set_error("comm_create: no current device"); return fail(MXDET_EHIP); }
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) {
    set_error("comm_create: cannot create the side stream"); return fail(MXDET_EHIP);
  }
  for (int i = 0; i < MXDET_COMM_MAX_INFLIGHT; ++i) {
    if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) != hipSuccess) {
      set_error("comm_create: cannot create events"); return fail(MXDET_EHIP);
    }
  }
  ncclUniqueId u;
  memcpy(&u, id, MXDET_COMM_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    set_error("comm_create: RCCL: %s", g_rccl.GetErrorString(r));
    c->comm = nullptr;
    return fail(MXDET_ERCCL);
  }
  *comm_out = c;
  return MXDET_OK;
}

extern "C" int mxdet_comm_destroy(mxdet_comm_t* c) {
  if (!c) return MXDET_OK;
  int rc = MXDET_OK;
  if (c->side) hipStreamSynchronize(c->side);
  if (c->comm && g_rccl.ok && g_rccl.CommDestroy(c->comm) != ncclSuccess) rc = MXDET_ERCCL;
  for (int i = 0; i < MXDET_COMM_MAX_INFLIGHT; ++i) {
    if (c->ready[i]) hipEventDestroy(c->ready[i]);
    if (c->done[i]) hipEventDestroy(c->done[i]);
  }
  if (c->side) hipStreamDestroy(c->side);
  delete c;
  return rc;
}

extern "C" int mxdet_allreduce_bucket(mxdet_comm_t* c, float* grad, int64_t count, mxdet_stream_t stream,
                                      int32_t* ticket_out) {
  clear_error();
  MXDET_REQUIRE(c && c->comm, MXDET_EINVAL, "allreduce_bucket: null communicator");
  MXDET_REQUIRE(grad != nullptr && count > 0, MXDET_EINVAL, "allreduce_bucket: empty bucket");
  const int slot = (int)(c->issued % MXDET_COMM_MAX_INFLIGHT);
  // the bucket is final once everything enqueued on the caller's stream so far has run
  MXDET_HIP(hipEventRecord(c->ready[slot], as_stream(stream)), "allreduce_bucket");
  MXDET_HIP(hipStreamWaitEvent(c->side, c->ready[slot], 0), "allreduce_bucket");
  MXDET_RCCL(g_rccl.AllReduce(grad, grad, (size_t)count, ncclFloat32, ncclSum, c->comm, c->side), "allreduce_bucket");
  MXDET_HIP(hipEventRecord(c->done[slot], c->side), "allreduce_bucket");
  if (ticket_out) *ticket_out = (int32_t)(c->issued & 0x7fffffff);
  ++c->issued;
  return MXDET_OK;
}

extern "C" int mxdet_comm_wait(mxdet_comm_t* c, int32_t ticket, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(c != nullptr, MXDET_EINVAL, "comm_wait: null communicator");
  if (c->issued == 0) return MXDET_OK;
  if (ticket < 0) {
    // collectives run in issue order on one stream: the youngest one's event covers all
    const int slot = (int)((c->issued - 1) % MXDET_COMM_MAX_INFLIGHT);
    MXDET_HIP(hipStreamWaitEvent(as_stream(stream), c->done[slot], 0), "comm_wait");
    return MXDET_OK;
  }
  // Event slot of the ticket (2^31 is a multiple of the ring size, so the wrapped ticket keeps its slot). If a younger
  // bucket has reused the slot, its event covers the older one too: collectives run in issue order on one stream.
  const int slot = (int)(ticket % MXDET_COMM_MAX_INFLIGHT);
  MXDET_HIP(hipStreamWaitEvent(as_stream(stream), c->done[slot], 0), "comm_wait");
  return MXDET_OK;
}

extern "C" int mxdet_comm_broadcast(mxdet_comm_t* c, void* buf, size_t bytes, int32_t root, mxdet_stream_t stream) {
  clear_error();
  MXDET_REQUIRE(c && c->comm, MXDET_EINVAL, "comm_broadcast: null communicator");
  MXDET_REQUIRE(buf != nullptr && bytes > 0, MXDET_EINVAL, "comm_broadcast: empty buffer");
  MXDET_REQUIRE(root >= 0 && root < c->world, MXDET_EINVAL, "comm_broadcast: root %d of %d", root, c->world);
  MXDET_RCCL(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, as_stream(stream)), "comm_broadcast");
  return MXDET_OK;
}
