// hsqp_comm_*: the batch axis over the GPUs of one node behind the C ABI (include/hsqp.h; SURVEY.md §8b "library owns ... RCCL comms", §8e:
// one process per GPU, contiguous blocks of ceil(B / world) instances, broadcast of the shared problem image + scatter of the shards before the
// solve, gather of the solutions after it, nothing inside the solve).  The reference has no counterpart: its solver runs one instance on the
// host (ocs2 SqpSolver behind /root/reference/humanoid_nmpc/humanoid_wb_mpc/src/WBMpcInterface.cpp:113-121); the Python host of this package
// does the same exchange through torch.distributed (wb_humanoid_mpc_amd/distributed.py), a C++ host has this.
//
// RCCL is bound at run time (dlopen of librccl.so, local scope): the solver library carries no link-time dependency on it — a single-GPU
// host never loads it, and a process that already holds a copy (PyTorch ships its own) is not handed a second set of global symbols.
// Scatter and gather are direct peer transfers (ncclSend / ncclRecv grouped on the root: xGMI is point to point, a shard is 2.4 MB at
// config 4), not ring collectives; the root's own block is a device-to-device copy.  Every call enqueues on the communicator's stream and
// returns after hipStreamSynchronize: the buffers may be handed to hsqp_upload_device / freed right away.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>

// RCCL's development header is used where the ROCm install has it; a ROCm without it still builds the solver library: the handful of
// declarations the exchange needs are restated below (the values are RCCL's public ABI, nccl.h: ncclChar = 0, ncclDouble = 8, ncclMax = 2,
// a 128-byte identifier), and the entry points are bound by name at run time either way.
#if defined(__has_include) && __has_include(<rccl/rccl.h>) && !defined(HSQP_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
const char* ncclGetErrorString(ncclResult_t);
ncclResult_t ncclBroadcast(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
}
#endif
static_assert((int)ncclChar == 0 && (int)ncclDouble == 8 && (int)ncclMax == 2 && NCCL_UNIQUE_ID_BYTES == 128, "RCCL's public constants");

#include "../../include/hsqp.h"

namespace {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  std::string err;
};

RcclApi& rccl() {
  static RcclApi api;
  return api;
}

// HSQP_RCCL_LIB names the library explicitly (then nothing else is tried: a test or a site that substitutes the transport means it);
// otherwise the soname, then the ROCm install.  One thread binds; concurrent first calls wait for it.
std::mutex& rccl_mutex() { static std::mutex m; return m; }
bool rccl_load() {
  std::lock_guard<std::mutex> lock(rccl_mutex());
  RcclApi& a = rccl();
  if (a.lib) return true;
  const char* env = getenv("HSQP_RCCL_LIB");
  const char* fallbacks[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  std::string why;
  auto try_open = [&](const char* n) {
    a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) { const char* de = dlerror(); why = de ? de : "dlopen failed"; }   // (dlerror() clears the message: read it once)
    return a.lib != nullptr;
  };
  if (env && *env) try_open(env);
  else for (const char* n : fallbacks) if (try_open(n)) break;
  if (!a.lib) { a.err = std::string("librccl not found (") + why + "); set HSQP_RCCL_LIB"; return false; }
  bool ok = true;
  auto sym = [&](auto& fn, const char* name) {
    fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(a.lib, name));
    if (!fn) { a.err = std::string("librccl lacks ") + name; ok = false; }
  };
  sym(a.GetUniqueId, "ncclGetUniqueId");
  sym(a.CommInitRank, "ncclCommInitRank");
  sym(a.CommDestroy, "ncclCommDestroy");
  sym(a.GetErrorString, "ncclGetErrorString");
  sym(a.Broadcast, "ncclBroadcast");
  sym(a.AllReduce, "ncclAllReduce");
  sym(a.Send, "ncclSend");
  sym(a.Recv, "ncclRecv");
  sym(a.GroupStart, "ncclGroupStart");
  sym(a.GroupEnd, "ncclGroupEnd");
  if (!ok) { dlclose(a.lib); a.lib = nullptr; }
  return ok;
}

thread_local std::string g_comm_create_error;

}  // namespace

struct hsqp_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* d_red = nullptr;        // staging of hsqp_comm_max
  int red_capacity = 0;
  std::string err;
};

static_assert(HSQP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the identifier the host ships between its processes is RCCL's");

namespace {

int fail(hsqp_comm* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
int hip_fail(hsqp_comm* c, hipError_t e, const char* what) { return fail(c, HSQP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
int nccl_fail(hsqp_comm* c, ncclResult_t r, const char* what) { return fail(c, HSQP_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(r)); }
#define COMM_HIP(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(c, e_, what); } while (0)
#define COMM_NCCL(expr, what) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return nccl_fail(c, r_, what); } while (0)

// block of rank r: [lo, hi)
void shard(int global_batch, int world, int r, int* lo, int* hi) {
  const int per = (global_batch + world - 1) / world;
  const int l = r * per < global_batch ? r * per : global_batch, h = l + per < global_batch ? l + per : global_batch;
  *lo = l; *hi = h;
}

}  // namespace

extern "C" {

int hsqp_comm_unique_id(void* id) {
  if (!id) return HSQP_ERR_BAD_ARG;
  if (!rccl_load()) { g_comm_create_error = rccl().err; return HSQP_ERR_HIP; }
  ncclUniqueId u;
  const ncclResult_t r = rccl().GetUniqueId(&u);
  if (r != ncclSuccess) { g_comm_create_error = std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r); return HSQP_ERR_HIP; }
  memcpy(id, &u, HSQP_COMM_ID_BYTES);
  return HSQP_OK;
}

const char* hsqp_comm_create_error(void) { return g_comm_create_error.c_str(); }

int hsqp_comm_create(hsqp_comm** out, const void* id, int rank, int world, int device) {
  if (out) *out = nullptr;
  if (!out || !id || world < 1 || rank < 0 || rank >= world || device < 0) { g_comm_create_error = "bad argument"; return HSQP_ERR_BAD_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_comm_create_error = "no HIP device (the comms move device buffers; there is no host path)"; return HSQP_ERR_NO_DEVICE; }
  if (device >= ndev) { g_comm_create_error = "device index out of range"; return HSQP_ERR_BAD_ARG; }
  if (!rccl_load()) { g_comm_create_error = rccl().err; return HSQP_ERR_HIP; }
  hsqp_comm* c = new hsqp_comm;
  c->rank = rank; c->world = world; c->device = device;
  auto bail = [&](const std::string& m) { g_comm_create_error = m; hsqp_comm_destroy(c); return HSQP_ERR_HIP; };
  if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice failed");
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate failed");
  ncclUniqueId u;
  memcpy(&u, id, HSQP_COMM_ID_BYTES);
  const ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) { c->comm = nullptr; return bail(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r)); }
  *out = c;
  return HSQP_OK;
}

void hsqp_comm_destroy(hsqp_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->d_red) (void)hipFree(c->d_red);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int hsqp_comm_rank(const hsqp_comm* c) { return c ? c->rank : -1; }
int hsqp_comm_world(const hsqp_comm* c) { return c ? c->world : 0; }
const char* hsqp_comm_last_error(const hsqp_comm* c) { return c ? c->err.c_str() : "null communicator"; }

int hsqp_comm_shard_of(int global_batch, int world, int rank, int* lo, int* hi) {
  if (!lo || !hi || global_batch < 0 || world < 1 || rank < 0 || rank >= world) return HSQP_ERR_BAD_ARG;
  shard(global_batch, world, rank, lo, hi);
  return HSQP_OK;
}

int hsqp_comm_shard(const hsqp_comm* c, int global_batch, int* lo, int* hi) {
  if (!c || !lo || !hi || global_batch < 0) return HSQP_ERR_BAD_ARG;
  shard(global_batch, c->world, c->rank, lo, hi);
  return HSQP_OK;
}

int hsqp_comm_broadcast(hsqp_comm* c, void* d_buf, long long bytes, int root) {
  if (!c || !d_buf || bytes < 0 || root < 0 || root >= c->world) return fail(c, HSQP_ERR_BAD_ARG, "hsqp_comm_broadcast: bad argument");
  COMM_HIP(hipSetDevice(c->device), "hipSetDevice");
  if (bytes > 0) COMM_NCCL(rccl().Broadcast(d_buf, d_buf, (size_t)bytes, ncclChar, root, c->comm, c->stream), "ncclBroadcast");
  COMM_HIP(hipStreamSynchronize(c->stream), "broadcast");
  return HSQP_OK;
}

// rows: the leading axis of the global array is the batch; a row is row_doubles contiguous doubles
int hsqp_comm_scatter_rows(hsqp_comm* c, const double* d_global, double* d_local, long long row_doubles, int global_batch, int root) {
  if (!c || !d_local || row_doubles < 1 || global_batch < 0 || root < 0 || root >= c->world || (c->rank == root && !d_global))
    return fail(c, HSQP_ERR_BAD_ARG, "hsqp_comm_scatter_rows: bad argument");
  COMM_HIP(hipSetDevice(c->device), "hipSetDevice");
  int lo, hi;
  shard(global_batch, c->world, c->rank, &lo, &hi);
  if (c->rank == root) {
    if (hi > lo) COMM_HIP(hipMemcpyAsync(d_local, d_global + (size_t)lo * row_doubles, (size_t)(hi - lo) * row_doubles * 8, hipMemcpyDeviceToDevice, c->stream), "scatter (own block)");
    if (c->world > 1) {
      COMM_NCCL(rccl().GroupStart(), "ncclGroupStart");
      for (int r = 0; r < c->world; ++r) {
        if (r == root) continue;
        int l, h;
        shard(global_batch, c->world, r, &l, &h);
        if (h > l) COMM_NCCL(rccl().Send(d_global + (size_t)l * row_doubles, (size_t)(h - l) * row_doubles, ncclDouble, r, c->comm, c->stream), "ncclSend");
      }
      COMM_NCCL(rccl().GroupEnd(), "ncclGroupEnd");
    }
  } else if (hi > lo) {
    COMM_NCCL(rccl().Recv(d_local, (size_t)(hi - lo) * row_doubles, ncclDouble, root, c->comm, c->stream), "ncclRecv");
  }
  COMM_HIP(hipStreamSynchronize(c->stream), "scatter");
  return HSQP_OK;
}

int hsqp_comm_gather_rows(hsqp_comm* c, const double* d_local, double* d_global, long long row_doubles, int global_batch, int root) {
  if (!c || !d_local || row_doubles < 1 || global_batch < 0 || root < 0 || root >= c->world || (c->rank == root && !d_global))
    return fail(c, HSQP_ERR_BAD_ARG, "hsqp_comm_gather_rows: bad argument");
  COMM_HIP(hipSetDevice(c->device), "hipSetDevice");
  int lo, hi;
  shard(global_batch, c->world, c->rank, &lo, &hi);
  if (c->rank == root) {
    if (hi > lo) COMM_HIP(hipMemcpyAsync(d_global + (size_t)lo * row_doubles, d_local, (size_t)(hi - lo) * row_doubles * 8, hipMemcpyDeviceToDevice, c->stream), "gather (own block)");
    if (c->world > 1) {
      COMM_NCCL(rccl().GroupStart(), "ncclGroupStart");
      for (int r = 0; r < c->world; ++r) {
        if (r == root) continue;
        int l, h;
        shard(global_batch, c->world, r, &l, &h);
        if (h > l) COMM_NCCL(rccl().Recv(d_global + (size_t)l * row_doubles, (size_t)(h - l) * row_doubles, ncclDouble, r, c->comm, c->stream), "ncclRecv");
      }
      COMM_NCCL(rccl().GroupEnd(), "ncclGroupEnd");
    }
  } else if (hi > lo) {
    COMM_NCCL(rccl().Send(d_local, (size_t)(hi - lo) * row_doubles, ncclDouble, root, c->comm, c->stream), "ncclSend");
  }
  COMM_HIP(hipStreamSynchronize(c->stream), "gather");
  return HSQP_OK;
}

// element-wise maximum over the ranks of n host values (the timing reduction of a benchmark: max over ranks of the elapsed time), in place
int hsqp_comm_max(hsqp_comm* c, double* values, int n) {
  if (!c || !values || n < 1) return fail(c, HSQP_ERR_BAD_ARG, "hsqp_comm_max: bad argument");
  COMM_HIP(hipSetDevice(c->device), "hipSetDevice");
  if (n > c->red_capacity) {
    if (c->d_red) (void)hipFree(c->d_red);
    c->d_red = nullptr; c->red_capacity = 0;
    COMM_HIP(hipMalloc(&c->d_red, (size_t)n * 8), "hipMalloc");
    c->red_capacity = n;
  }
  COMM_HIP(hipMemcpyAsync(c->d_red, values, (size_t)n * 8, hipMemcpyHostToDevice, c->stream), "upload");
  COMM_NCCL(rccl().AllReduce(c->d_red, c->d_red, (size_t)n, ncclDouble, ncclMax, c->comm, c->stream), "ncclAllReduce");
  COMM_HIP(hipMemcpyAsync(values, c->d_red, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream), "download");
  COMM_HIP(hipStreamSynchronize(c->stream), "max");
  return HSQP_OK;
}

int hsqp_comm_barrier(hsqp_comm* c) {
  double one = 1.0;
  return hsqp_comm_max(c, &one, 1);
}

}  // extern "C"
