// The one collective of the path behind the C ABI: the framebuffer-shard gather over RCCL / xGMI; no reference counterpart
// (the reference renders on one GPU).  librccl is opened lazily: a single-GPU host never needs it.
//
// Communication model: every rank's shard (pt_local_shard: maxTilesPerRank x 1024 float4, 16.6 MB per peer for a 4K image on 8 GPUs) goes to
// the root in ONE grouped operation -- nranks-1 ncclRecv on the root, one ncclSend per peer.  xGMI is point-to-point, so each peer's shard
// travels on its own link and the gather is bounded by one link (~153 GB/s): ~0.1-0.3 ms.  The root then places the tiles (pt_scatter_shards).
// Works with one process per GPU (pt_comm_get_unique_id + pt_comm_init_rank) and with one process driving all GPUs (pt_comm_init_all;
// bracket the per-context pt_gather_shards calls with pt_comm_group_begin / pt_comm_group_end).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include "../../include/pt_api.h"

extern "C" int  pt_comm_internal_shard(pt_context* ctx, void** shard, size_t* bytes, int* rank, int* nranks, void** gatherBuf, hipStream_t* stream, int* device);
extern "C" void pt_comm_internal_fail(pt_context* ctx, int code, const char* msg);

namespace {
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId)    GetUniqueId    = nullptr;
  decltype(&ncclCommInitRank)   CommInitRank   = nullptr;
  decltype(&ncclCommInitAll)    CommInitAll    = nullptr;
  decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
  decltype(&ncclCommCount)      CommCount      = nullptr;
  decltype(&ncclCommUserRank)   CommUserRank   = nullptr;
  decltype(&ncclGroupStart)     GroupStart     = nullptr;
  decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
  decltype(&ncclSend)           Send           = nullptr;
  decltype(&ncclRecv)           Recv           = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string                   err;
};
Rccl&      rccl() { static Rccl r; return r; }
std::mutex g_lock;

bool load()
{
  std::lock_guard<std::mutex> g(g_lock);
  Rccl&                       r = rccl();
  if(r.handle)
    return true;
  for(const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if((r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
      break;
  if(!r.handle)
  {
    r.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
    return false;
  }
#define SYM(n)                                                                 \
  if(!(r.n = reinterpret_cast<decltype(r.n)>(dlsym(r.handle, "nccl" #n))))     \
  {                                                                            \
    r.err    = "librccl.so lacks nccl" #n;                                     \
    r.handle = nullptr;                                                        \
    return false;                                                              \
  }
  SYM(GetUniqueId) SYM(CommInitRank) SYM(CommInitAll) SYM(CommDestroy) SYM(CommCount) SYM(CommUserRank) SYM(GroupStart) SYM(GroupEnd) SYM(Send) SYM(Recv) SYM(GetErrorString)
#undef SYM
  return true;
}
int nccl_fail(pt_context* ctx, const char* what, ncclResult_t rc)
{
  char msg[256];
  std::snprintf(msg, sizeof(msg), "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
  if(ctx)
    pt_comm_internal_fail(ctx, PT_ERR_HIP, msg);
  return PT_ERR_HIP;
}
}  // namespace

extern "C" {

int pt_comm_get_unique_id(unsigned char id_out[PT_COMM_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "ncclUniqueId size");
  if(!id_out || !load())
    return PT_ERR_INVALID;
  ncclUniqueId id;
  if(rccl().GetUniqueId(&id) != ncclSuccess)
    return PT_ERR_HIP;
  std::memcpy(id_out, &id, sizeof(id));
  return PT_OK;
}

int pt_comm_init_rank(int nranks, const unsigned char id[PT_COMM_ID_BYTES], int rank, int device_ordinal, pt_comm** out_comm)
{
  if(!out_comm || !id || nranks < 1 || rank < 0 || rank >= nranks || !load())
    return PT_ERR_INVALID;
  if(hipSetDevice(device_ordinal) != hipSuccess)
    return PT_ERR_NO_DEVICE;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  if(rccl().CommInitRank(&c, nranks, uid, rank) != ncclSuccess)
    return PT_ERR_HIP;
  *out_comm = reinterpret_cast<pt_comm*>(c);
  return PT_OK;
}

int pt_comm_init_all(int ndev, const int* device_ordinals, pt_comm** out_comms)
{
  if(!out_comms || ndev < 1 || !load())
    return PT_ERR_INVALID;
  if(rccl().CommInitAll(reinterpret_cast<ncclComm_t*>(out_comms), ndev, device_ordinals) != ncclSuccess)
    return PT_ERR_HIP;
  return PT_OK;
}

int pt_comm_destroy(pt_comm* comm)
{
  if(!comm || !load())
    return PT_ERR_INVALID;
  return rccl().CommDestroy(reinterpret_cast<ncclComm_t>(comm)) == ncclSuccess ? PT_OK : PT_ERR_HIP;
}

int pt_comm_group_begin(void) { return load() && rccl().GroupStart() == ncclSuccess ? PT_OK : PT_ERR_HIP; }
int pt_comm_group_end(void) { return load() && rccl().GroupEnd() == ncclSuccess ? PT_OK : PT_ERR_HIP; }

// Enqueues this context's part of the gather on its stream.  Root: receives every peer's shard into its gather buffer (its own shard is a
// device copy); others: send.  Call pt_gather_finish on the root afterwards (it waits and places the tiles).
int pt_gather_shards(pt_context* ctx, pt_comm* comm, int root)
{
  if(!ctx)
    return PT_ERR_INVALID;
  if(!comm || !load())
  {
    pt_comm_internal_fail(ctx, PT_ERR_INVALID, rccl().err.empty() ? "pt_gather_shards: null communicator" : rccl().err.c_str());
    return PT_ERR_INVALID;
  }
  void*       shard = nullptr;
  void*       gbuf  = nullptr;
  size_t      bytes = 0;
  int         rank = 0, nranks = 1, device = 0;
  hipStream_t stream = nullptr;
  int         rc     = pt_comm_internal_shard(ctx, &shard, &bytes, &rank, &nranks, root >= 0 ? &gbuf : nullptr, &stream, &device);
  if(rc != PT_OK)
    return rc;
  ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
  int        cn = 0, cr = -1;
  if(rccl().CommCount(c, &cn) != ncclSuccess || rccl().CommUserRank(c, &cr) != ncclSuccess || cn != nranks || cr != rank || root < 0 || root >= nranks)
  {
    pt_comm_internal_fail(ctx, PT_ERR_INVALID, "pt_gather_shards: the communicator's rank / size differ from pt_set_shard's, or bad root");
    return PT_ERR_INVALID;
  }
  if(hipSetDevice(device) != hipSuccess)
    return PT_ERR_HIP;
  const size_t count = bytes / sizeof(float);
  ncclResult_t r;
  if((r = rccl().GroupStart()) != ncclSuccess)
    return nccl_fail(ctx, "ncclGroupStart", r);
  if(rank == root)
  {
    for(int p = 0; p < nranks; ++p)
    {
      char* dst = static_cast<char*>(gbuf) + size_t(p) * bytes;
      if(p == rank)
      {
        if(hipMemcpyAsync(dst, shard, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
          return PT_ERR_HIP;
      }
      else if((r = rccl().Recv(dst, count, ncclFloat, p, c, stream)) != ncclSuccess)
        return nccl_fail(ctx, "ncclRecv", r);
    }
  }
  else if((r = rccl().Send(shard, count, ncclFloat, root, c, stream)) != ncclSuccess)
    return nccl_fail(ctx, "ncclSend", r);
  if((r = rccl().GroupEnd()) != ncclSuccess)
    return nccl_fail(ctx, "ncclGroupEnd", r);
  return PT_OK;
}
}
