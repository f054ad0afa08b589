// comm.hip -- collectives behind the C ABI: RCCL over xGMI, called from C++ (no Python, no torch).
//
// The reference is ONE process whose operators run on scheduler workers (scheduler/operator_task.cpp:163-200,
// abstract_scheduler.cpp:53-63).  Its multi-GPU shape is therefore one process, one worker thread per GPU:
//   hy_bind_device(d)      the calling thread works on device d from now on (its stream, buffer pool, scratch arena and every
//                          column it creates belong to that device; threads that never call it use hy_init's device)
//   hy_comm_init_all       one communicator per device of the process (ncclCommInitAll): RCCL's single-process mode
//   hy_comm_init_rank      one communicator per PROCESS (ncclCommInitRank with an id from hy_comm_unique_id): the one-process-per-GPU
//                          launch of bench.py, should a C++ host ever use it
//   hy_comm_all_reduce / all_gather / all_to_all_v   on the calling thread's stream, asynchronous like a kernel launch
//   hy_comm_group_begin / end   ncclGroupStart / ncclGroupEnd: one thread that drives several devices' communicators issues
//                          each collective for all of them inside one group (the single-process rule of NCCL / RCCL)
// What the sharded operators exchange: fixed-slot partial aggregates (all_reduce), (key, partial) tables
// and build-side columns (all_gather), (key, RowID) tuples by key % G (all_to_all_v = grouped ncclSend / ncclRecv, every pair of
// GPUs over its own xGMI link).  librccl is opened when the first communicator is created, not when this library is loaded:
// a single-GPU user never touches it, and a process that already holds an RCCL (PyTorch's) shares that copy.
#include "hy_device.hpp"

#include <atomic>
#include <string>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace hy { struct LocalExchange; }

struct hy_comm {
  ncclComm_t comm = nullptr;                       // RCCL, or ...
  std::shared_ptr<hy::LocalExchange> local;        // ... ranks that share ONE device (RCCL refuses those): copies inside its HBM
  int device = 0;
  uint32_t world = 1, rank = 0;
  uint64_t collectives = 0;                        // this rank's collectives over `local` so far (LocalExchange::failed)
};

namespace hy {

void bind_thread_to(int device);   // runtime.hip

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommInitAll) comm_init_all = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;
static std::string g_rccl_why;   // why librccl is not usable, captured once (dlerror() clears itself when read)

static hy_status rccl(Rccl** out) {
  std::call_once(g_rccl_once, [] {
    void* handle = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already holds (PyTorch's)
      if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!handle) {
      const char* why = dlerror();
      g_rccl_why = why ? why : "dlopen failed";
      return;
    }
    Rccl r;
    r.handle = handle;
#define HY_RCCL_SYMBOL(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(handle, name))
    HY_RCCL_SYMBOL(get_unique_id, "ncclGetUniqueId");
    HY_RCCL_SYMBOL(comm_init_rank, "ncclCommInitRank");
    HY_RCCL_SYMBOL(comm_init_all, "ncclCommInitAll");
    HY_RCCL_SYMBOL(comm_destroy, "ncclCommDestroy");
    HY_RCCL_SYMBOL(all_reduce, "ncclAllReduce");
    HY_RCCL_SYMBOL(all_gather, "ncclAllGather");
    HY_RCCL_SYMBOL(send, "ncclSend");
    HY_RCCL_SYMBOL(recv, "ncclRecv");
    HY_RCCL_SYMBOL(group_start, "ncclGroupStart");
    HY_RCCL_SYMBOL(group_end, "ncclGroupEnd");
    HY_RCCL_SYMBOL(error_string, "ncclGetErrorString");
#undef HY_RCCL_SYMBOL
    if (r.get_unique_id && r.comm_init_rank && r.comm_init_all && r.comm_destroy && r.all_reduce && r.all_gather && r.send && r.recv && r.group_start && r.group_end && r.error_string)
      g_rccl = r;
    else
      g_rccl_why = "a collective's symbol is missing";
  });
  if (!g_rccl.handle) return fail(HY_ERR_DEVICE, "librccl.so could not be loaded (%s): the collectives of the multi-GPU path need RCCL", g_rccl_why.c_str());
  *out = &g_rccl;
  return HY_OK;
}

#define HY_RCCL(r, expr)                                                                                                      \
  do {                                                                                                                        \
    const ncclResult_t res__ = (expr);                                                                                        \
    if (res__ != ncclSuccess) return ::hy::fail(HY_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, (r)->error_string(res__), __FILE__, __LINE__); \
  } while (0)

static bool nccl_type(uint32_t data_type, ncclDataType_t* out, size_t* width) {
  switch (data_type) {
    case HY_TYPE_INT: *out = ncclInt32; *width = 4; return true;
    case HY_TYPE_LONG: *out = ncclInt64; *width = 8; return true;
    case HY_TYPE_FLOAT: *out = ncclFloat32; *width = 4; return true;
    case HY_TYPE_DOUBLE: *out = ncclFloat64; *width = 8; return true;
    default: return false;
  }
}


// ---- ranks on one device ---------------------------------------------------------------------------------------------------------
// ncclCommInitAll rejects a device list that names one GPU twice ("invalid usage").  Worker threads that share a GPU -- two Hyrise
// scheduler workers on one device, or the two-rank tests on a one-GPU box -- exchange through that GPU's memory instead: every rank
// publishes its send buffer, all ranks meet, every rank copies what is addressed to it (device-to-device on its own stream), all
// ranks meet again.  Reductions of such ranks are staged through the host (they are a few cells per group).  Same entry points,
// same results; nothing here touches a second device.
struct LocalExchange {
  explicit LocalExchange(uint32_t n) : world(n), send(n, nullptr), send_bytes(n, nullptr) {}
  void meet() {
    std::unique_lock<std::mutex> lock(mutex);
    const uint64_t generation = round;
    if (++arrived == world) {
      arrived = 0;
      ++round;
      everyone.notify_all();
    } else {
      everyone.wait(lock, [&] { return round != generation; });
    }
  }
  const uint32_t world;
  std::mutex mutex;
  std::condition_variable everyone;
  uint32_t arrived = 0;
  uint64_t round = 0;
  std::vector<const void*> send;
  std::vector<const uint64_t*> send_bytes;
  // A rank that fails (its stream, an argument) still takes part in both meetings of a collective and raises the collective's flag before the
  // first: the others skip the exchange and report the failure too, instead of waiting for the rank that left.  Two flags, taken in turn
  // (every rank counts its collectives: hy_comm::collectives): collective k uses failed[k & 1], and rank 0 clears the OTHER one between k's
  // two meetings -- nobody can be in collective k + 1 yet, and everybody has long finished reading k - 1's.  (One flag cleared after the
  // last meeting could erase what a faster rank had already raised for the next collective.)
  std::atomic<uint32_t> failed[2] = {{0}, {0}};
};

template <typename T>
static void reduce_cells(T* into, const T* from, uint64_t count, uint32_t op) {
  for (uint64_t i = 0; i < count; ++i) into[i] = op == HY_COMM_SUM ? static_cast<T>(into[i] + from[i]) : op == HY_COMM_MIN ? std::min(into[i], from[i]) : std::max(into[i], from[i]);
}

static hy_status local_all_reduce(hy_comm* comm, const void* send, void* recv, uint64_t count, uint32_t data_type, size_t width, uint32_t op) {
  LocalExchange& x = *comm->local;
  hipStream_t stream = current_stream();
  hy_status status = HY_OK;
  std::atomic<uint32_t>& failed = x.failed[comm->collectives & 1];
  std::atomic<uint32_t>& next_failed = x.failed[(comm->collectives + 1) & 1];
  ++comm->collectives;
  if (hipStreamSynchronize(stream) != hipSuccess) { status = fail(HY_ERR_DEVICE, "hy_comm_all_reduce: this rank's stream failed"); failed.store(1); }
  x.send[comm->rank] = send;
  x.meet();
  if (comm->rank == 0) next_failed.store(0);
  if (status == HY_OK && failed.load()) status = fail(HY_ERR_DEVICE, "hy_comm_all_reduce: a co-located rank failed");
  std::vector<unsigned char> total(count * width), piece(count * width);
  for (uint32_t peer = 0; peer < x.world && status == HY_OK; ++peer) {
    if (hipMemcpyAsync(peer == 0 ? total.data() : piece.data(), x.send[peer], count * width, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
      status = fail(HY_ERR_DEVICE, "hy_comm_all_reduce: reading a co-located rank's cells failed");
      break;
    }
    if (peer == 0) continue;
    switch (data_type) {
      case HY_TYPE_INT: reduce_cells(reinterpret_cast<int32_t*>(total.data()), reinterpret_cast<const int32_t*>(piece.data()), count, op); break;
      case HY_TYPE_LONG: reduce_cells(reinterpret_cast<int64_t*>(total.data()), reinterpret_cast<const int64_t*>(piece.data()), count, op); break;
      case HY_TYPE_FLOAT: reduce_cells(reinterpret_cast<float*>(total.data()), reinterpret_cast<const float*>(piece.data()), count, op); break;
      default: reduce_cells(reinterpret_cast<double*>(total.data()), reinterpret_cast<const double*>(piece.data()), count, op); break;
    }
  }
  x.meet();   // everyone has read every send buffer: recv may alias send
  if (status == HY_OK && count && (hipMemcpyAsync(recv, total.data(), count * width, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess))
    status = fail(HY_ERR_DEVICE, "hy_comm_all_reduce: writing the reduced cells failed");
  return status;
}

// peer p's bytes for this rank start at from_peer(p) inside p's send buffer and are `bytes(p)` long; they land back to back in recv
template <typename Offset, typename Bytes>
static hy_status local_collect(hy_comm* comm, const void* send, const uint64_t* send_bytes, void* recv, Offset from_peer, Bytes bytes) {
  LocalExchange& x = *comm->local;
  hipStream_t stream = current_stream();
  hy_status status = HY_OK;
  std::atomic<uint32_t>& failed = x.failed[comm->collectives & 1];
  std::atomic<uint32_t>& next_failed = x.failed[(comm->collectives + 1) & 1];
  ++comm->collectives;
  if (hipStreamSynchronize(stream) != hipSuccess) { status = fail(HY_ERR_DEVICE, "this rank's stream failed before a collective"); failed.store(1); }
  x.send[comm->rank] = send;
  x.send_bytes[comm->rank] = send_bytes;
  x.meet();
  if (comm->rank == 0) next_failed.store(0);
  if (status == HY_OK && failed.load()) status = fail(HY_ERR_DEVICE, "a co-located rank failed before a collective");
  uint64_t at = 0;
  for (uint32_t peer = 0; peer < x.world; ++peer) {
    const uint64_t n = bytes(peer);
    if (n && status == HY_OK && hipMemcpyAsync(static_cast<char*>(recv) + at, static_cast<const char*>(x.send[peer]) + from_peer(peer), n, hipMemcpyDeviceToDevice, stream) != hipSuccess)
      status = fail(HY_ERR_DEVICE, "copying a co-located rank's bytes failed");
    at += n;
  }
  if (hipStreamSynchronize(stream) != hipSuccess && status == HY_OK) status = fail(HY_ERR_DEVICE, "copying a co-located rank's bytes failed");
  x.meet();   // the send buffers (and the callers' send_bytes arrays) may go now
  return status;
}

}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_bind_device(int32_t device) {
  int n = 0;
  HY_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(HY_ERR_INVALID, "hy_bind_device: device %d out of range [0,%d)", device, n);
  bind_thread_to(device);
  return HY_OK;
}

hy_status hy_comm_unique_id(void* id_128_bytes) {
  if (!id_128_bytes) return fail(HY_ERR_INVALID, "hy_comm_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == HY_COMM_ID_BYTES, "hy_comm_unique_id hands out ncclUniqueId");
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  HY_RCCL(r, r->get_unique_id(static_cast<ncclUniqueId*>(id_128_bytes)));
  return HY_OK;
}

hy_status hy_comm_init_rank(const void* id_128_bytes, uint32_t world, uint32_t rank, hy_comm** out) {
  if (!id_128_bytes || !out || world == 0 || rank >= world) return fail(HY_ERR_INVALID, "hy_comm_init_rank: invalid argument");
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  (void)current_stream();   // binds the thread to its device
  auto comm = new hy_comm();
  (void)hipGetDevice(&comm->device);
  comm->world = world;
  comm->rank = rank;
  ncclUniqueId id;
  std::memcpy(&id, id_128_bytes, sizeof(id));
  const ncclResult_t res = r->comm_init_rank(&comm->comm, static_cast<int>(world), id, static_cast<int>(rank));
  if (res != ncclSuccess) { delete comm; return fail(HY_ERR_DEVICE, "ncclCommInitRank failed: %s", r->error_string(res)); }
  *out = comm;
  return HY_OK;
}

hy_status hy_comm_init_all(const int32_t* devices, uint32_t n_devices, hy_comm** out) {
  if (!devices || !out || n_devices == 0) return fail(HY_ERR_INVALID, "hy_comm_init_all: invalid argument");
  bool one_device = n_devices > 1;
  for (uint32_t i = 1; i < n_devices; ++i) one_device &= devices[i] == devices[0];
  if (one_device) {   // (a list that names a device twice next to other devices stays RCCL's to refuse)
    auto exchange = std::make_shared<LocalExchange>(n_devices);
    for (uint32_t i = 0; i < n_devices; ++i) {
      out[i] = new hy_comm();
      out[i]->local = exchange;
      out[i]->device = devices[0];
      out[i]->world = n_devices;
      out[i]->rank = i;
    }
    return HY_OK;
  }
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  std::vector<ncclComm_t> comms(n_devices);
  std::vector<int> list(devices, devices + n_devices);
  int before = 0;
  (void)hipGetDevice(&before);
  const ncclResult_t res = r->comm_init_all(comms.data(), static_cast<int>(n_devices), list.data());
  (void)hipSetDevice(before);
  if (res != ncclSuccess) return fail(HY_ERR_DEVICE, "ncclCommInitAll failed: %s", r->error_string(res));
  for (uint32_t i = 0; i < n_devices; ++i) {
    out[i] = new hy_comm();
    out[i]->comm = comms[i];
    out[i]->device = devices[i];
    out[i]->world = n_devices;
    out[i]->rank = i;
  }
  return HY_OK;
}

hy_status hy_comm_destroy(hy_comm* comm) {
  if (!comm) return HY_OK;
  Rccl* r = nullptr;
  if (comm->comm && rccl(&r) == HY_OK) (void)r->comm_destroy(comm->comm);
  delete comm;
  return HY_OK;
}

hy_status hy_comm_rank(const hy_comm* comm, uint32_t* rank, uint32_t* world) {
  if (!comm || !rank || !world) return fail(HY_ERR_INVALID, "hy_comm_rank: null argument");
  *rank = comm->rank;
  *world = comm->world;
  return HY_OK;
}

hy_status hy_comm_group_begin(void) {
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  HY_RCCL(r, r->group_start());
  return HY_OK;
}

hy_status hy_comm_group_end(void) {
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  HY_RCCL(r, r->group_end());
  return HY_OK;
}

hy_status hy_comm_all_reduce(hy_comm* comm, const void* send, void* recv, uint64_t count, uint32_t data_type, uint32_t op) {
  if (!comm) return fail(HY_ERR_INVALID, "hy_comm_all_reduce: null argument");
  ncclDataType_t type = ncclInt8;
  size_t width = 0;
  const bool bad = (count && (!send || !recv)) || !nccl_type(data_type, &type, &width) || op > HY_COMM_MAX;
  if (bad && comm->local) {   // (the co-located ranks are waiting at the meeting points: go there, with the failure flag up)
    const uint64_t k = comm->collectives++;
    comm->local->failed[k & 1].store(1);
    comm->local->meet();
    if (comm->rank == 0) comm->local->failed[(k + 1) & 1].store(0);
    comm->local->meet();
  }
  if (bad) return fail(HY_ERR_INVALID, "hy_comm_all_reduce: null argument, data type %u or operation %u", data_type, op);
  if (comm->local) return local_all_reduce(comm, send, recv, count, data_type, width, op);
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  const ncclRedOp_t reduce = op == HY_COMM_SUM ? ncclSum : op == HY_COMM_MIN ? ncclMin : ncclMax;
  HY_RCCL(r, r->all_reduce(send, recv, count, type, reduce, comm->comm, current_stream()));
  return HY_OK;
}

hy_status hy_comm_all_gather(hy_comm* comm, const void* send, void* recv, uint64_t bytes_per_rank) {
  if (!comm || (bytes_per_rank && (!send || !recv))) return fail(HY_ERR_INVALID, "hy_comm_all_gather: null argument");
  if (comm->local) return local_collect(comm, send, nullptr, recv, [](uint32_t) { return uint64_t{0}; }, [&](uint32_t) { return bytes_per_rank; });
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  HY_RCCL(r, r->all_gather(send, recv, bytes_per_rank, ncclInt8, comm->comm, current_stream()));
  return HY_OK;
}

// send: the bytes for rank 0, then for rank 1, ... (send_bytes[g] each, back to back); recv likewise (recv_bytes[g] from rank g).
// One group of ncclSend / ncclRecv: every pair of GPUs exchanges over its own xGMI link at the same time.
hy_status hy_comm_all_to_all_v(hy_comm* comm, const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes) {
  if (!comm || !send_bytes || !recv_bytes) return fail(HY_ERR_INVALID, "hy_comm_all_to_all_v: null argument");
  if (comm->local) {
    LocalExchange& x = *comm->local;
    const uint32_t me = comm->rank;
    return local_collect(comm, send, send_bytes, recv,
                         [&](uint32_t peer) { uint64_t at = 0; for (uint32_t q = 0; q < me; ++q) at += x.send_bytes[peer][q]; return at; },
                         [&](uint32_t peer) { return recv_bytes[peer]; });
  }
  Rccl* r = nullptr;
  HY_TRY(rccl(&r));
  hipStream_t stream = current_stream();
  HY_RCCL(r, r->group_start());
  // (a send or recv that fails must not leave the thread's group open -- later collectives would queue behind it and never start:
  //  the group is always closed, the first error is reported)
  ncclResult_t first_error = ncclSuccess;
  uint64_t send_at = 0, recv_at = 0;
  for (uint32_t peer = 0; peer < comm->world && first_error == ncclSuccess; ++peer) {
    if (send_bytes[peer]) first_error = r->send(static_cast<const char*>(send) + send_at, send_bytes[peer], ncclInt8, static_cast<int>(peer), comm->comm, stream);
    if (recv_bytes[peer] && first_error == ncclSuccess) first_error = r->recv(static_cast<char*>(recv) + recv_at, recv_bytes[peer], ncclInt8, static_cast<int>(peer), comm->comm, stream);
    send_at += send_bytes[peer];
    recv_at += recv_bytes[peer];
  }
  const ncclResult_t closed = r->group_end();
  if (first_error == ncclSuccess) first_error = closed;
  if (first_error != ncclSuccess) return fail(HY_ERR_DEVICE, "hy_comm_all_to_all_v failed: %s", r->error_string(first_error));
  return HY_OK;
}

}  // extern "C"
