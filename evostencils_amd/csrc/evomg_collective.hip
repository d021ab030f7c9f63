// evomg_collective.hip — the population's ONE exchange step behind the C-ABI: every rank (one process per GPU) contributes its
// candidates' fitness rows and receives everybody's, over RCCL / xGMI, without torch.
//
// Reference counterpart: the mpi4py object collectives of evostencils/optimization/program.py — `allgather` of
// (index, fitness) pairs after the round-robin re-evaluation (:534-536) and the per-generation `allgather(offspring)` (:582),
// helpers :285-310. SURVEY.md 8(b) lists `evomg_gather_fitness` as part of the boundary; 8(e): "one collective per generation:
// ncclAllGather (or ncclGather to rank 0) of [n_local x {rho, t_ms, iters}] f64". Payload: a few KiB — latency bound, the link
// count of the xGMI mesh is irrelevant; there is no collective anywhere on the data path of a solve.
//
// librccl is opened on first use (dlopen "librccl.so.1": 570 MB that a process which never leaves its GPU should not map);
// the communicator's unique id travels through a small file (rank 0 writes it atomically, the others poll), so no MPI, no
// torch and no socket code is needed here. Every call returns a status, never aborts, and never blocks without a bound
// before the communicator exists (afterwards RCCL's own semantics apply: a collective completes when all ranks entered it).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <ctime>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "../../include/evomg.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x): kept local so that the build does not depend on the header
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                  // ncclSuccess == 0
enum { kNcclFloat64 = 8 };                 // ncclDataType_t: ncclFloat64 = ncclDouble = 8

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string error;
};

Rccl load_rccl();
// loaded once, by whichever thread asks first (a function-local static is initialised under the runtime's own guard)
Rccl& rccl() {
  static Rccl r = load_rccl();
  return r;
}
Rccl load_rccl() {
  Rccl r;
  // The RCCL that belongs to the HIP runtime THIS process runs on: a process that imported torch first runs on the runtime
  // torch bundles (and torch's librccl, built for it, answers to the same SONAME), any other process on /opt/rocm's. Mixing
  // them fails in ncclCommInitRank ("unhandled cuda error": torch's RCCL 2.26 on the system's HIP 7.2, seen on the GPU box), so
  // the library next to the loaded libamdhip64 is tried first, by path (a path is not matched against SONAMEs already loaded).
  std::string beside_hip, beside_hip2;
  Dl_info hip_info;
  if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &hip_info) && hip_info.dli_fname) {
    const std::string path(hip_info.dli_fname);
    const size_t slash = path.rfind('/');
    if (slash != std::string::npos) {
      beside_hip = path.substr(0, slash) + "/librccl.so.1";
      beside_hip2 = path.substr(0, slash) + "/librccl.so";
    }
  }
  const char* names[] = {getenv("EVOMG_RCCL_LIB"), beside_hip.c_str(), beside_hip2.c_str(), "librccl.so.1", "librccl.so",
                         "/opt/rocm/lib/librccl.so.1"};
  // EVOMG_RCCL_ONLY=<path>: that library or none (a deployment that pins its RCCL; the test of the not-found path)
  const char* only = getenv("EVOMG_RCCL_ONLY");
  if (only && *only) r.lib = dlopen(only, RTLD_NOW | RTLD_LOCAL);
  else
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
  if (!r.lib) {
    const char* why = dlerror();                   // one call: it hands the message over and clears it
    r.error = std::string("librccl not found: ") + (why ? why : "");
    return r;
  }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(r.lib, name);
    if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
    return p;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
  return r;
}

thread_local char g_comm_error[512] = "";

constexpr char kMagic[8] = {'E', 'V', 'O', 'M', 'G', 'I', 'D', '2'};
// Which launch an id file belongs to (the file's age says nothing — a rank that starts a minute late must still
// accept rank 0's id, a relaunch seconds after a crash must not accept the dead job's): a nonce every rank of ONE launch
// computes alike and no other launch can — the launcher process the ranks are children of (its pid AND its start time, from
// /proc: pids are reused) together with the rendezvous the launcher handed down (TORCHELASTIC_RUN_ID, MASTER_PORT,
// WORLD_SIZE). Rank 0 writes it behind the magic; a reader takes a file with another nonce as absent and keeps waiting (a
// stale id would send it into ncclCommInitRank with nobody on the other side, which has no timeout).
unsigned long long launch_nonce() {
  unsigned long long h = 1469598103934665603ULL;                // FNV-1a
  auto mix = [&](const char* s) {
    for (; s && *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ULL; }
    h ^= 0xff; h *= 1099511628211ULL;
  };
  const long ppid = (long)getppid();
  char buf[1024];
  snprintf(buf, sizeof buf, "%ld", ppid);
  mix(buf);
  snprintf(buf, sizeof buf, "/proc/%ld/stat", ppid);
  if (FILE* f = fopen(buf, "r")) {
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    // field 22 (starttime) counted from behind the command name, which may itself contain blanks and parentheses
    const char* q = strrchr(buf, ')');
    int field = 2;
    for (; q && *q; ++q)
      if (*q == ' ' && ++field == 22) { mix(q + 1 < buf + n ? std::string(q + 1, strcspn(q + 1, " ")).c_str() : ""); break; }
  }
  mix(getenv("TORCHELASTIC_RUN_ID"));
  mix(getenv("MASTER_PORT"));
  mix(getenv("WORLD_SIZE"));
  return h;
}

}  // namespace

struct evomg_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* dsend = nullptr;
  double* drecv = nullptr;
  size_t cap_rows = 0;             // rows per rank the device buffers hold
  std::string id_path;
  char err[512] = "";
};

#define COMM_HIP_TRY(c, expr)                                                                                   \
  do {                                                                                                          \
    hipError_t e_ = (expr);                                                                                     \
    if (e_ != hipSuccess) {                                                                                     \
      snprintf((c)->err, sizeof (c)->err, "%s failed: %s", #expr, hipGetErrorString(e_));                       \
      return EVOMG_ERR_DEVICE;                                                                                  \
    }                                                                                                           \
  } while (0)
#define COMM_NCCL_TRY(c, expr)                                                                                  \
  do {                                                                                                          \
    ncclResult_t r_ = (expr);                                                                                   \
    if (r_ != 0) {                                                                                              \
      snprintf((c)->err, sizeof (c)->err, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
      return EVOMG_ERR_DEVICE;                                                                                  \
    }                                                                                                           \
  } while (0)

extern "C" {

#pragma GCC visibility push(default)

const char* evomg_comm_last_error(const evomg_comm_t* c) { return c ? c->err : g_comm_error; }

int evomg_rccl_version(void) {
  Rccl& r = rccl();
  int v = 0;
  if (!r.error.empty() || !r.GetVersion || r.GetVersion(&v) != 0) return 0;
  return v;
}

int evomg_comm_create(int rank, int world, int device_id, const char* id_path, int timeout_ms, evomg_comm_t** out) {
  if (!out) return EVOMG_ERR_INVALID;
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || !id_path || !*id_path) {
    snprintf(g_comm_error, sizeof g_comm_error, "comm_create: bad rank / world / id path");
    return EVOMG_ERR_INVALID;
  }
  Rccl& r = rccl();
  if (!r.error.empty()) {
    snprintf(g_comm_error, sizeof g_comm_error, "%s", r.error.c_str());
    return EVOMG_ERR_UNSUPPORTED;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) {
    snprintf(g_comm_error, sizeof g_comm_error, "comm_create: device %d not available (%d devices)", device_id, count);
    return EVOMG_ERR_DEVICE;
  }
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  const std::string path(id_path);
  if (rank == 0) {
    (void)unlink(path.c_str());                    // whatever an earlier job left under this name
    if (r.GetUniqueId(&id) != 0) {
      snprintf(g_comm_error, sizeof g_comm_error, "ncclGetUniqueId failed");
      return EVOMG_ERR_DEVICE;
    }
    // atomically: a reader sees either no file or the whole id
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    const unsigned long long nonce = launch_nonce();
    if (!f || fwrite(kMagic, 1, sizeof kMagic, f) != sizeof kMagic || fwrite(&nonce, 1, sizeof nonce, f) != sizeof nonce ||
        fwrite(&id, 1, sizeof id, f) != sizeof id) {
      if (f) fclose(f);
      snprintf(g_comm_error, sizeof g_comm_error, "comm_create: cannot write %s", tmp.c_str());
      return EVOMG_ERR_INVALID;
    }
    fclose(f);
    if (rename(tmp.c_str(), path.c_str()) != 0) {
      snprintf(g_comm_error, sizeof g_comm_error, "comm_create: cannot publish %s", path.c_str());
      return EVOMG_ERR_INVALID;
    }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    bool have = false, saw_foreign = false;
    while (!have) {
      FILE* f = fopen(path.c_str(), "rb");
      if (f) {
        char magic[sizeof kMagic];
        unsigned long long nonce = 0;
        have = fread(magic, 1, sizeof magic, f) == sizeof magic && memcmp(magic, kMagic, sizeof magic) == 0 &&
               fread(&nonce, 1, sizeof nonce, f) == sizeof nonce && nonce == launch_nonce() &&
               fread(&id, 1, sizeof id, f) == sizeof id;
        if (!have) saw_foreign = true;
        fclose(f);
      }
      if (have) break;
      const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (timeout_ms >= 0 && waited > timeout_ms) {
        snprintf(g_comm_error, sizeof g_comm_error, "comm_create: rank %d waited %d ms for %s%s", rank, timeout_ms, path.c_str(),
                 saw_foreign ? " (a file is there, but written by another launch: different launcher process or rendezvous)" : "");
        return EVOMG_ERR_DEVICE;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
  }
  evomg_comm* c = new evomg_comm;
  c->rank = rank;
  c->world = world;
  c->device = device_id;
  c->id_path = path;
  auto fail = [&](int rc) {
    snprintf(g_comm_error, sizeof g_comm_error, "%s", c->err);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (rank == 0) (void)unlink(path.c_str());     // nobody may join an id whose root never came up
    delete c;
    return rc;
  };
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    snprintf(c->err, sizeof c->err, "comm_create: %s", hipGetErrorString(e));
    return fail(EVOMG_ERR_DEVICE);
  }
  const ncclResult_t nr = r.CommInitRank(&c->comm, world, id, rank);
  if (nr != 0) {
    snprintf(c->err, sizeof c->err, "ncclCommInitRank failed: %s", r.GetErrorString ? r.GetErrorString(nr) : "?");
    return fail(EVOMG_ERR_DEVICE);
  }
  *out = c;
  return EVOMG_OK;
}

void evomg_comm_destroy(evomg_comm_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  if (c->dsend) (void)hipFree(c->dsend);
  if (c->drecv) (void)hipFree(c->drecv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->rank == 0) (void)unlink(c->id_path.c_str());
  delete c;
}

int evomg_allgather_fitness(evomg_comm_t* c, const double* local, int n_local, double* all) {
  if (!c || n_local < 0 || (n_local > 0 && (!local || !all))) return EVOMG_ERR_INVALID;
  if (n_local == 0) return EVOMG_OK;
  COMM_HIP_TRY(c, hipSetDevice(c->device));
  if ((size_t)n_local > c->cap_rows) {
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    c->dsend = c->drecv = nullptr;
    c->cap_rows = 0;
    COMM_HIP_TRY(c, hipMalloc(&c->dsend, sizeof(double) * EVOMG_FITNESS_COLUMNS * n_local));
    COMM_HIP_TRY(c, hipMalloc(&c->drecv, sizeof(double) * EVOMG_FITNESS_COLUMNS * n_local * c->world));
    c->cap_rows = (size_t)n_local;
  }
  const size_t count = (size_t)EVOMG_FITNESS_COLUMNS * n_local;
  COMM_HIP_TRY(c, hipMemcpyAsync(c->dsend, local, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  COMM_NCCL_TRY(c, rccl().AllGather(c->dsend, c->drecv, count, kNcclFloat64, c->comm, c->stream));
  COMM_HIP_TRY(c, hipMemcpyAsync(all, c->drecv, sizeof(double) * count * c->world, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EVOMG_OK;
}

int evomg_gather_fitness(evomg_comm_t* c, const double* local, int n_local, double* all, int root) {
  if (!c || root < 0 || root >= c->world || n_local < 0 || (n_local > 0 && !local)) return EVOMG_ERR_INVALID;
  if (n_local == 0) return EVOMG_OK;
  if (c->rank == root && !all) return EVOMG_ERR_INVALID;
  // a few KiB: the all-gather IS the gather (one ring step more than a rooted gather would need, far below the launch
  // latency of the collective); ranks other than the root simply do not copy the table back
  COMM_HIP_TRY(c, hipSetDevice(c->device));
  if ((size_t)n_local > c->cap_rows) {
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    c->dsend = c->drecv = nullptr;
    c->cap_rows = 0;
    COMM_HIP_TRY(c, hipMalloc(&c->dsend, sizeof(double) * EVOMG_FITNESS_COLUMNS * n_local));
    COMM_HIP_TRY(c, hipMalloc(&c->drecv, sizeof(double) * EVOMG_FITNESS_COLUMNS * n_local * c->world));
    c->cap_rows = (size_t)n_local;
  }
  const size_t count = (size_t)EVOMG_FITNESS_COLUMNS * n_local;
  COMM_HIP_TRY(c, hipMemcpyAsync(c->dsend, local, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  COMM_NCCL_TRY(c, rccl().AllGather(c->dsend, c->drecv, count, kNcclFloat64, c->comm, c->stream));
  if (c->rank == root) COMM_HIP_TRY(c, hipMemcpyAsync(all, c->drecv, sizeof(double) * count * c->world, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return EVOMG_OK;
}

#pragma GCC visibility pop

}  // extern "C"
