// The one collective of the path behind the C ABI: the framebuffer-shard gather over RCCL / xGMI; no reference counterpart
// (the reference renders on one GPU).  librccl is opened lazily: a single-GPU host never needs it, and libptmi.so builds without the RCCL
// development headers -- the handful of types and entry points used here are declared locally (they are RCCL's stable C ABI, identical to
// NCCL's: ncclUniqueId is 128 opaque bytes, ncclComm_t an opaque pointer, ncclResult_t / ncclDataType_t plain enums).
//
// Communication model: every rank's shard (pt_local_shard: maxTilesPerRank x 1024 float4, 16.6 MB per peer for a 4K image on 8 GPUs) goes to
// the root in ONE grouped operation -- nranks-1 ncclRecv on the root, one ncclSend per peer.  xGMI is point-to-point, so each peer's shard
// travels on its own link and the gather is bounded by one link (~153 GB/s): ~0.1-0.3 ms.  The root then places the tiles (pt_scatter_shards).
// Works with one process per GPU (pt_comm_get_unique_id + pt_comm_init_rank) and with one process driving all GPUs (pt_comm_init_all;
// bracket the per-context pt_gather_shards calls with pt_comm_group_begin / pt_comm_group_end).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include "../../include/pt_api.h"

extern "C" int  pt_comm_internal_shard(pt_context* ctx, void** shard, size_t* bytes, int* rank, int* nranks, int root, void** gatherBuf, hipStream_t* stream, int* device);
extern "C" void pt_comm_internal_fail(pt_context* ctx, int code, const char* msg);

namespace {
// RCCL's C ABI, as far as this file uses it
typedef struct { char internal[PT_COMM_ID_BYTES]; } rcclUniqueId;
typedef void* rcclComm_t;
typedef int   rcclResult_t;  // ncclSuccess == 0
constexpr int kRcclFloat = 7; // ncclFloat32

struct Rccl {
  void* handle = nullptr;
  rcclResult_t (*GetUniqueId)(rcclUniqueId*)                                             = nullptr;
  rcclResult_t (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int)                      = nullptr;
  rcclResult_t (*CommInitAll)(rcclComm_t*, int, const int*)                              = nullptr;
  rcclResult_t (*CommDestroy)(rcclComm_t)                                                = nullptr;
  rcclResult_t (*CommCount)(const rcclComm_t, int*)                                      = nullptr;
  rcclResult_t (*CommUserRank)(const rcclComm_t, int*)                                   = nullptr;
  rcclResult_t (*GroupStart)()                                                           = nullptr;
  rcclResult_t (*GroupEnd)()                                                             = nullptr;
  rcclResult_t (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t)           = nullptr;
  rcclResult_t (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t)                 = nullptr;
  const char* (*GetErrorString)(rcclResult_t)                                            = nullptr;
  rcclResult_t (*GetVersion)(int*)                                                       = nullptr;  // optional
  std::string err;  // why the library is unavailable / the last failure
};
Rccl&      rccl() { static Rccl r; return r; }
std::mutex g_lock;

bool load()
{
  std::lock_guard<std::mutex> g(g_lock);
  Rccl&                       r = rccl();
  if(r.handle)
    return true;
  // PT_RCCL_LIB: test hook (a name that cannot be opened exercises the "library missing" answers on a host that has RCCL)
  const char* forced = std::getenv("PT_RCCL_LIB");
  void*       h      = nullptr;
  std::string why;
  for(const char* name : {forced ? forced : "librccl.so.1", forced ? forced : "librccl.so", forced ? forced : "/opt/rocm/lib/librccl.so.1"})
  {
    // RTLD_DEEPBIND: RCCL must talk to the HIP runtime this library is linked against (its own DT_NEEDED libamdhip64), not to a second runtime a
    // host process may have loaded globally (PyTorch wheels bundle one): device pointers and streams of one runtime mean nothing to the other
    if((h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND)))
      break;
    const char* e = dlerror();  // read ONCE: glibc clears the message on read
    if(why.empty())
      why = e ? e : "dlopen failed";
  }
  if(!h)
  {
    r.err = "librccl.so not found: " + why;
    return false;
  }
#define SYM(n)                                                                  \
  if(!(r.n = reinterpret_cast<decltype(r.n)>(dlsym(h, "nccl" #n))))             \
  {                                                                             \
    r.err = "librccl.so lacks nccl" #n;                                         \
    dlclose(h);                                                                 \
    return false;                                                               \
  }
  SYM(GetUniqueId) SYM(CommInitRank) SYM(CommInitAll) SYM(CommDestroy) SYM(CommCount) SYM(CommUserRank) SYM(GroupStart) SYM(GroupEnd) SYM(Send) SYM(Recv) SYM(GetErrorString)
#undef SYM
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
  r.handle = h;
  r.err.clear();
  return true;
}
int fail(int code, const char* what, rcclResult_t rc = 0)
{
  std::lock_guard<std::mutex> g(g_lock);
  Rccl&                       r = rccl();
  r.err = what;
  if(rc != 0)
    r.err += std::string(": ") + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
  return code;
}
int nccl_fail(pt_context* ctx, const char* what, rcclResult_t rc)
{
  fail(PT_ERR_HIP, what, rc);
  if(ctx)
    pt_comm_internal_fail(ctx, PT_ERR_HIP, rccl().err.c_str());
  return PT_ERR_HIP;
}
}  // namespace

extern "C" {

const char* pt_comm_last_error(void)
{
  static thread_local std::string copy;
  std::lock_guard<std::mutex>     g(g_lock);
  copy = rccl().err;
  return copy.c_str();
}

int pt_comm_get_unique_id(unsigned char id_out[PT_COMM_ID_BYTES])
{
  static_assert(sizeof(rcclUniqueId) == PT_COMM_ID_BYTES, "ncclUniqueId size");
  if(!id_out)
    return fail(PT_ERR_INVALID, "pt_comm_get_unique_id: null output");
  if(!load())
    return PT_ERR_UNAVAILABLE;
  rcclUniqueId id;
  rcclResult_t r;
  if((r = rccl().GetUniqueId(&id)) != 0)
    return fail(PT_ERR_HIP, "ncclGetUniqueId", r);
  std::memcpy(id_out, &id, sizeof(id));
  return PT_OK;
}

int pt_comm_init_rank(int nranks, const unsigned char id[PT_COMM_ID_BYTES], int rank, int device_ordinal, pt_comm** out_comm)
{
  if(!out_comm || !id || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(PT_ERR_INVALID, "pt_comm_init_rank: bad argument");
  if(!load())
    return PT_ERR_UNAVAILABLE;
  if(hipSetDevice(device_ordinal) != hipSuccess)
    return fail(PT_ERR_NO_DEVICE, "pt_comm_init_rank: no such device");
  rcclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  rcclComm_t   c = nullptr;
  rcclResult_t r;
  (void)hipGetLastError();  // RCCL reads the thread's sticky HIP error: a stale hipErrorNotReady of an event query must not become its "unhandled cuda error"
  if((r = rccl().CommInitRank(&c, nranks, uid, rank)) != 0)
    return fail(PT_ERR_HIP, "ncclCommInitRank", r);
  *out_comm = reinterpret_cast<pt_comm*>(c);
  return PT_OK;
}

int pt_comm_init_all(int ndev, const int* device_ordinals, pt_comm** out_comms)
{
  if(!out_comms || ndev < 1)
    return fail(PT_ERR_INVALID, "pt_comm_init_all: bad argument");
  if(!load())
    return PT_ERR_UNAVAILABLE;
  int have = 0;
  if(hipGetDeviceCount(&have) != hipSuccess || have < ndev)
  {
    char msg[128];
    std::snprintf(msg, sizeof(msg), "pt_comm_init_all: %d devices requested, %d visible", ndev, have);
    return fail(PT_ERR_NO_DEVICE, msg);
  }
  rcclResult_t r;
  (void)hipGetLastError();  // (see pt_comm_init_rank)
  if((r = rccl().CommInitAll(reinterpret_cast<rcclComm_t*>(out_comms), ndev, device_ordinals)) != 0)
    return fail(PT_ERR_HIP, "ncclCommInitAll", r);
  return PT_OK;
}

int pt_comm_destroy(pt_comm* comm)
{
  if(!comm)
    return fail(PT_ERR_INVALID, "pt_comm_destroy: null communicator");
  if(!load())
    return PT_ERR_UNAVAILABLE;
  rcclResult_t r = rccl().CommDestroy(reinterpret_cast<rcclComm_t>(comm));
  return r == 0 ? PT_OK : fail(PT_ERR_HIP, "ncclCommDestroy", r);
}

int pt_comm_count(pt_comm* comm, int* out_nranks)
{
  if(!comm || !out_nranks)
    return fail(PT_ERR_INVALID, "pt_comm_count: null argument");
  if(!load())
    return PT_ERR_UNAVAILABLE;
  rcclResult_t r = rccl().CommCount(reinterpret_cast<rcclComm_t>(comm), out_nranks);
  return r == 0 ? PT_OK : fail(PT_ERR_HIP, "ncclCommCount", r);
}

int pt_comm_version(int* out_version)
{
  if(!out_version)
    return fail(PT_ERR_INVALID, "pt_comm_version: null argument");
  if(!load())
    return PT_ERR_UNAVAILABLE;
  *out_version = 0;
  if(!rccl().GetVersion)
    return PT_OK;  // (an RCCL without ncclGetVersion: 0)
  rcclResult_t r = rccl().GetVersion(out_version);
  return r == 0 ? PT_OK : fail(PT_ERR_HIP, "ncclGetVersion", r);
}

int pt_comm_group_begin(void)
{
  if(!load())
    return PT_ERR_UNAVAILABLE;
  rcclResult_t r = rccl().GroupStart();
  return r == 0 ? PT_OK : fail(PT_ERR_HIP, "ncclGroupStart", r);
}
int pt_comm_group_end(void)
{
  if(!load())
    return PT_ERR_UNAVAILABLE;
  rcclResult_t r = rccl().GroupEnd();
  return r == 0 ? PT_OK : fail(PT_ERR_HIP, "ncclGroupEnd", r);
}

// Enqueues this context's part of the gather on its stream.  Root: receives every peer's shard into its gather buffer (its own shard is a
// device copy); others: send.  Call pt_gather_finish on the root afterwards (it waits and places the tiles).
// An RCCL group that was opened is ALWAYS closed, also on an error in between: a thread left inside an open group would queue every later
// RCCL call (including the caller's own pt_comm_group_end / pt_comm_destroy) and the peers would block in their matching Send / Recv.
int pt_gather_shards(pt_context* ctx, pt_comm* comm, int root)
{
  if(!ctx)
    return PT_ERR_INVALID;
  if(!comm)
  {
    pt_comm_internal_fail(ctx, PT_ERR_INVALID, "pt_gather_shards: null communicator");
    return PT_ERR_INVALID;
  }
  if(!load())
  {
    pt_comm_internal_fail(ctx, PT_ERR_UNAVAILABLE, pt_comm_last_error());
    return PT_ERR_UNAVAILABLE;
  }
  void*       shard = nullptr;
  void*       gbuf  = nullptr;
  size_t      bytes = 0;
  int         rank = 0, nranks = 1, device = 0;
  hipStream_t stream = nullptr;
  // validates root against pt_set_shard's nranks and allocates the gather buffer on the root only
  int         rc     = pt_comm_internal_shard(ctx, &shard, &bytes, &rank, &nranks, root, &gbuf, &stream, &device);
  if(rc != PT_OK)
    return rc;
  rcclComm_t c  = reinterpret_cast<rcclComm_t>(comm);
  int        cn = 0, cr = -1;
  if(rccl().CommCount(c, &cn) != 0 || rccl().CommUserRank(c, &cr) != 0 || cn != nranks || cr != rank)
  {
    pt_comm_internal_fail(ctx, PT_ERR_INVALID, "pt_gather_shards: the communicator's rank / size differ from pt_set_shard's");
    return PT_ERR_INVALID;
  }
  if(hipSetDevice(device) != hipSuccess)
  {  // every error return after pt_comm_internal_shard goes through pt_comm_internal_fail: a later pt_gather_finish must see "nothing enqueued"
    pt_comm_internal_fail(ctx, PT_ERR_HIP, "pt_gather_shards: hipSetDevice failed");
    return PT_ERR_HIP;
  }
  const size_t count = bytes / sizeof(float);
  rcclResult_t r;
  (void)hipGetLastError();  // (see pt_comm_init_rank)
  if((r = rccl().GroupStart()) != 0)
    return nccl_fail(ctx, "ncclGroupStart", r);
  int          firstErr = PT_OK;
  const char*  firstWhat = nullptr;
  rcclResult_t firstRc = 0;
  if(rank == root)
  {
    for(int p = 0; p < nranks && firstErr == PT_OK; ++p)
    {
      char* dst = static_cast<char*>(gbuf) + size_t(p) * bytes;
      if(p == rank)
      {
        if(hipMemcpyAsync(dst, shard, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        {
          firstErr  = PT_ERR_HIP;
          firstWhat = "hipMemcpyAsync of the root's own shard";
        }
      }
      else if((r = rccl().Recv(dst, count, kRcclFloat, p, c, stream)) != 0)
      {
        firstErr  = PT_ERR_HIP;
        firstWhat = "ncclRecv";
        firstRc   = r;
      }
    }
  }
  else if((r = rccl().Send(shard, count, kRcclFloat, root, c, stream)) != 0)
  {
    firstErr  = PT_ERR_HIP;
    firstWhat = "ncclSend";
    firstRc   = r;
  }
  r = rccl().GroupEnd();  // always: see above
  if(firstErr != PT_OK)
    return nccl_fail(ctx, firstWhat, firstRc);
  if(r != 0)
    return nccl_fail(ctx, "ncclGroupEnd", r);
  return PT_OK;
}
}
