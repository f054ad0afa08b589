// runtime.hip -- process/device runtime behind the C ABI: device selection, thread-local stream + error text,
// the per-thread scratch arena, and the residency cache (hy_column) that makes encoded segments device-visible once.
// One process per GPU (hy_init(device)); every ABI call is thread-safe and re-entrant: the only shared mutable state
// is thread-local (reference threading contract: scan_chunk is invoked concurrently from many workers,
// table_scan.cpp:129-131 -- here one call covers all chunks, and concurrent calls come from different threads).
#include "hy_device.hpp"
#include <algorithm>

#include <atomic>
#include <cstring>
#include <thread>

namespace hy {

static thread_local std::string t_error;
// The options of include/hyrise_amd.h (HY_OPT_*) with their defaults; hy_set_option stores, the operators load (hy_options.hpp).
std::atomic<int64_t> g_options[HY_OPT_COUNT];
namespace {
struct OptionDefaults {
  OptionDefaults() {
    for (auto& v : g_options) v.store(0, std::memory_order_relaxed);
    auto set = [](uint32_t id, int64_t v) { g_options[id].store(v, std::memory_order_relaxed); };
    set(HY_OPT_JOIN_RANK_TABLE, 1);
    set(HY_OPT_JOIN_HINT, 1);
    set(HY_OPT_JOIN_PKFK, 1);
    set(HY_OPT_JOIN_LDS_BUILD, 1);
    set(HY_OPT_JOIN_LDS_BUILD_TILES, 2048);
    set(HY_OPT_JOIN_FILL_WGS_PER_CU, 4);
    set(HY_OPT_JOIN_HAND_OVER_RANKS, 1 << 20);
    set(HY_OPT_AGG_SPILL_SHIFT, 3);
    set(HY_OPT_AGG_SMALL_DOMAIN, 1);
    set(HY_OPT_FUSED_SMALL_DOMAIN, 1);
    set(HY_OPT_SCAN_TWO_COLUMNS, 1);
    set(HY_OPT_STAR_FUSED_PROBE, 1);
    set(HY_OPT_STAR_FUSED_FINISH, 1);
  }
};
OptionDefaults g_option_defaults;   // (static initialisation: before any entry point can run)
}  // namespace

static thread_local hipStream_t t_stream = nullptr;
static thread_local Scratch t_scratch;

hy_status fail(hy_status code, const char* fmt, ...) {
  char buffer[1024];
  va_list args;
  va_start(args, fmt);
  vsnprintf(buffer, sizeof(buffer), fmt, args);
  va_end(args);
  t_error = buffer;
  return code;
}

// The device hy_init chose, for every thread of the process: hipSetDevice only binds the CALLING thread, and the operators
// run on Hyrise's scheduler workers (abstract_scheduler.cpp:53-63), not on the thread that loaded the plugin.  Every thread
// binds itself the first time it reaches the library (all entry points pass through current_stream / the pools below).
static std::atomic<int> g_device{-1};
static thread_local int t_bound_device = -1;
static thread_local int t_own_device = -1;   // hy_bind_device: this thread's device (one worker thread per GPU in a multi-GPU process), else hy_init's
void bind_thread_device() {
  const int device = t_own_device >= 0 ? t_own_device : g_device.load(std::memory_order_acquire);
  if (device >= 0 && t_bound_device != device) {
    (void)hipSetDevice(device);
    t_bound_device = device;
  }
}

// hy_bind_device (comm.hip): everything thread-local -- stream, buffer pool, scratch arena, pinned staging -- was created on the device the
// thread was bound to, so a thread may only change its device while it holds none of it (a fresh worker, or after hy_shutdown).
hy_status on_this_device(const hy_column* column, const char* entry_point) {
  if (!column) return HY_OK;
  bind_thread_device();
  const int here = t_bound_device >= 0 ? t_bound_device : 0;
  if (column->device != here) return fail(HY_ERR_INVALID, "%s: the column lives on device %d, the calling thread works on device %d (hy_bind_device)", entry_point, column->device, here);
  return on_this_device(column->ref, entry_point);
}

void bind_thread_to(int device) {
  t_own_device = device;
  bind_thread_device();
}

hipStream_t current_stream() {
  bind_thread_device();
  return t_stream;
}

struct Profile {
  bool enabled = false;
  uint32_t period = 1;              // every period-th bracket of a kernel is timed (timed launches cost ~7 us of stream time each)
  uint32_t counter[HY_KERNEL_KINDS] = {};
  bool open = false;                // profile_begin recorded, profile_end pending
  std::vector<hipEvent_t> events;   // start/stop pairs, reused across profiling sessions
  std::vector<uint8_t> kinds;       // [pair] which kernel the pair brackets (HY_KERNEL_*)
  size_t used = 0;
};
static thread_local Profile t_profile;

static hipEvent_t next_event() {
  Profile& p = t_profile;
  if (p.used == p.events.size()) {
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    p.events.push_back(e);
  }
  return p.events[p.used++];
}

static bool sampled(uint32_t kind) {
  Profile& p = t_profile;
  if (!p.enabled || p.used >= 16384) return false;
  if (kind >= HY_KERNEL_KINDS) kind = 0;
  if (p.counter[kind]++ % p.period != 0) return false;
  if (p.kinds.size() <= p.used / 2) p.kinds.resize(p.used / 2 + 1);
  p.kinds[p.used / 2] = static_cast<uint8_t>(kind);
  return true;
}

void profile_begin(hipStream_t stream, uint32_t kind) {
  t_profile.open = sampled(kind);
  if (t_profile.open) (void)hipEventRecord(next_event(), stream);
}

void profile_end(hipStream_t stream) {
  if (!t_profile.open) return;
  t_profile.open = false;
  (void)hipEventRecord(next_event(), stream);
}

// A start/stop pair for ONE kernel: handed to hipExtLaunchKernelGGL, which stamps them from the dispatch packet itself
// (no extra barrier packets on the stream, unlike hipEventRecord before and after the launch).
bool profile_events(hipEvent_t* start, hipEvent_t* stop, uint32_t kind) {
  *start = *stop = nullptr;
  if (!sampled(kind)) return false;
  *start = next_event();
  *stop = next_event();
  return true;
}


struct BufferPool {
  std::vector<std::pair<size_t, void*>> free_blocks;
  ~BufferPool() { for (auto& b : free_blocks) (void)hipFree(b.second); }
};
static thread_local BufferPool t_pool;

hy_status pool_acquire(size_t bytes, void** ptr, size_t* capacity) {
  bind_thread_device();
  size_t rounded = 4096;
  while (rounded < bytes) rounded <<= 1;
  for (size_t i = 0; i < t_pool.free_blocks.size(); ++i) {
    if (t_pool.free_blocks[i].first == rounded) {
      *ptr = t_pool.free_blocks[i].second;
      *capacity = rounded;
      t_pool.free_blocks.erase(t_pool.free_blocks.begin() + i);
      return HY_OK;
    }
  }
  hipError_t err = hipMalloc(ptr, rounded);
  if (err != hipSuccess) {   // release the pool and retry once
    for (auto& b : t_pool.free_blocks) (void)hipFree(b.second);
    t_pool.free_blocks.clear();
    err = hipMalloc(ptr, rounded);
  }
  if (err != hipSuccess) { *ptr = nullptr; return fail(HY_ERR_DEVICE, "hipMalloc(%zu) failed: %s", rounded, hipGetErrorString(err)); }
  *capacity = rounded;
  return HY_OK;
}

void pool_release(void* ptr, size_t capacity) {
  if (ptr) t_pool.free_blocks.emplace_back(capacity, ptr);
}

hy_status DeviceBuffer::alloc(size_t bytes) {
  if (ptr) {   // a buffer that is sized again gives its block back first
    if (!borrowed) pool_release(ptr, capacity);
    ptr = nullptr;
    capacity = 0;
  }
  borrowed = false;
  return pool_acquire(bytes, &ptr, &capacity);
}

void DeviceBuffer::borrow(void* block) {
  if (ptr && !borrowed) pool_release(ptr, capacity);
  ptr = block;
  capacity = 0;
  borrowed = true;
}

DeviceBuffer::~DeviceBuffer() { if (!borrowed) pool_release(ptr, capacity); }

struct PinnedStaging {
  void* host = nullptr;
  void* device = nullptr;
  size_t bytes = 0;
  ~PinnedStaging() { if (host) (void)hipHostFree(host); }
};
static thread_local PinnedStaging t_staging;

// Descriptor tables of columns over DEVICE memory (operator results: a reference column per operator and column of a plan) travel through
// a ring of pinned slots: the copy is queued on the thread's stream and nobody waits for it -- the slot is written again only after its
// event has passed (sixteen uploads later).  A synchronise per created column was 20 - 30 us of idle device, a dozen times per SSB join.
struct DescriptorRing {
  static constexpr uint32_t SLOTS = 16;
  static constexpr size_t SLOT_BYTES = size_t{1} << 19;
  unsigned char* host = nullptr;
  hipEvent_t sent[SLOTS] = {};
  bool busy[SLOTS] = {};
  uint32_t next = 0;
  void release() {
    for (uint32_t i = 0; i < SLOTS; ++i) {
      if (sent[i]) (void)hipEventDestroy(sent[i]);
      sent[i] = nullptr;
      busy[i] = false;
    }
    if (host) (void)hipHostFree(host);
    host = nullptr;
  }
};
static thread_local DescriptorRing t_descriptor_ring;

hy_status pinned_staging(size_t bytes, void** host, void** device) {
  bind_thread_device();
  if (bytes > t_staging.bytes) {
    if (t_staging.host) (void)hipHostFree(t_staging.host);
    t_staging.host = t_staging.device = nullptr;
    t_staging.bytes = 0;
    size_t rounded = 1 << 16;
    while (rounded < bytes) rounded <<= 1;
    HY_HIP(hipHostMalloc(&t_staging.host, rounded, hipHostMallocMapped));
    HY_HIP(hipHostGetDevicePointer(&t_staging.device, t_staging.host, 0));
    t_staging.bytes = rounded;
  }
  *host = t_staging.host;
  *device = t_staging.device;
  return HY_OK;
}

Scratch& scratch() {
  bind_thread_device();
  return t_scratch;
}

hy_status Scratch::reserve(size_t bytes) {
  bytes = align_up(bytes + 4096, 1 << 20);
  if (!ticket) {
    HY_HIP(hipMalloc(reinterpret_cast<void**>(&ticket), 256));
    HY_HIP(hipMemset(ticket, 0, 256));
    ticket_base = 0;
    epoch = 0;
  }
  if (bytes > capacity) {
    if (base) {
      HY_HIP(hipStreamSynchronize(t_stream));
      HY_HIP(hipFree(base));
      base = nullptr;
      capacity = 0;
    }
    HY_HIP(hipMalloc(&base, bytes));
    capacity = bytes;
  }
  used = 0;
  return HY_OK;
}

// Starts a new look-back epoch.  Status words are tagged {state:2, epoch:30, value:32}; the buffer holds nothing but
// status words, so anything left from earlier launches carries an older epoch and reads as "not yet published".
// The ticket counter is never reset either: the launch subtracts ticket_base.
hy_status Scratch::begin_launch(size_t status_words, uint32_t tickets) {
  if (status_words > status_capacity) {
    if (status) {
      HY_HIP(hipStreamSynchronize(t_stream));
      HY_HIP(hipFree(status));
      status = nullptr;
    }
    status_capacity = align_up(status_words + 1024, 4096);
    HY_HIP(hipMalloc(reinterpret_cast<void**>(&status), status_capacity * sizeof(uint64_t)));
    HY_HIP(hipMemsetAsync(status, 0, status_capacity * sizeof(uint64_t), t_stream));
  }
  if (epoch >= (1u << 30) - 2 || ticket_base > 0xF0000000u - tickets) {
    HY_HIP(hipMemsetAsync(status, 0, status_capacity * sizeof(uint64_t), t_stream));
    HY_HIP(hipMemsetAsync(ticket, 0, 256, t_stream));
    epoch = 0;
    ticket_base = 0;
  }
  ++epoch;
  return HY_OK;
}

void* Scratch::carve(size_t bytes) {
  const size_t offset = align_up(used, 256);
  if (offset + bytes > capacity) return nullptr;
  used = offset + bytes;
  return static_cast<char*>(base) + offset;
}

}  // namespace hy

using namespace hy;

extern "C" {

int32_t hy_abi_version(void) { return HY_ABI_VERSION; }

const char* hy_last_error(void) { return t_error.c_str(); }

hy_status hy_device_count(int32_t* count) {
  if (!count) return fail(HY_ERR_INVALID, "hy_device_count: null argument");
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) {
    *count = 0;
    return fail(HY_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(err));
  }
  *count = n;
  return HY_OK;
}

hy_status hy_set_option(uint32_t option_id, int64_t value) {
  if (option_id >= HY_OPT_COUNT) return fail(HY_ERR_INVALID, "hy_set_option: no option %u", option_id);
  g_options[option_id].store(value, std::memory_order_relaxed);
  return HY_OK;
}

hy_status hy_get_option(uint32_t option_id, int64_t* value) {
  if (option_id >= HY_OPT_COUNT || !value) return fail(HY_ERR_INVALID, "hy_get_option: no option %u", option_id);
  *value = option(option_id);
  return HY_OK;
}

hy_status hy_init(int32_t device) {
  int n = 0;
  HY_HIP(hipGetDeviceCount(&n));
  if (n <= 0) return fail(HY_ERR_DEVICE, "hy_init: no HIP device visible -- the MI355X path has no CPU fallback");
  if (device < 0 || device >= n) return fail(HY_ERR_INVALID, "hy_init: device %d out of range [0,%d)", device, n);
  HY_HIP(hipSetDevice(device));
  g_device.store(device, std::memory_order_release);   // (every other thread binds itself on its first call: bind_thread_device)
  t_bound_device = device;
  hipDeviceProp_t prop;
  HY_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !option(HY_OPT_ALLOW_ANY_ARCH)) {
    return fail(HY_ERR_DEVICE, "hy_init: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                prop.gcnArchName);
  }
  return HY_OK;
}

hy_status hy_set_profiling(int32_t enabled) {
  t_profile.enabled = enabled > 0;
  t_profile.period = enabled > 1 ? static_cast<uint32_t>(enabled) : 1;
  for (uint32_t& c : t_profile.counter) c = 0;
  t_profile.open = false;
  t_profile.used = 0;
  return HY_OK;
}

// Sum and count of the recorded pairs of one kernel kind (HY_KERNEL_KINDS: all of them); waits for the events.
static hy_status profile_sum(uint32_t kind, float* total_milliseconds, uint32_t* launches) {
  Profile& p = t_profile;
  float total = 0.f;
  uint32_t count = 0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    if (kind != HY_KERNEL_KINDS && (i / 2 >= p.kinds.size() || p.kinds[i / 2] != kind)) continue;
    float ms = 0.f;
    HY_HIP(hipEventSynchronize(p.events[i + 1]));
    HY_HIP(hipEventElapsedTime(&ms, p.events[i], p.events[i + 1]));
    total += ms;
    ++count;
  }
  *total_milliseconds = total;
  *launches = count;
  return HY_OK;
}

hy_status hy_profile_read(float* total_milliseconds, uint32_t* launches) {
  if (!total_milliseconds || !launches) return fail(HY_ERR_INVALID, "hy_profile_read: null argument");
  HY_TRY(profile_sum(HY_KERNEL_KINDS, total_milliseconds, launches));
  t_profile.used = 0;
  return HY_OK;
}

// What an event pair around a kernel measures beyond the kernel: the pair is stamped from the dispatch packet (before the first
// wave starts, after the completion signal).  Measured once per process on an empty kernel (median of 32 launches); a profiler's
// begin / end timestamps of the same kernel are shorter by about this much.
__global__ void profile_empty_kernel() {}
hy_status hy_profile_event_overhead(float* milliseconds) {
  if (!milliseconds) return fail(HY_ERR_INVALID, "hy_profile_event_overhead: null argument");
  static std::atomic<uint32_t> cached{0};   // float bits; 0 = not measured yet
  uint32_t bits = cached.load(std::memory_order_acquire);
  if (!bits) {
    hipStream_t stream = current_stream();
    hipEvent_t start = nullptr, stop = nullptr;
    HY_HIP(hipEventCreate(&start));
    HY_HIP(hipEventCreate(&stop));
    std::vector<float> samples;
    for (int i = 0; i < 40; ++i) {
      hipExtLaunchKernelGGL(profile_empty_kernel, dim3(1), dim3(64), 0, stream, start, stop, 0);
      HY_HIP(hipEventSynchronize(stop));
      float ms = 0.f;
      HY_HIP(hipEventElapsedTime(&ms, start, stop));
      if (i >= 8) samples.push_back(ms);
    }
    (void)hipEventDestroy(start);
    (void)hipEventDestroy(stop);
    std::sort(samples.begin(), samples.end());
    float median = samples[samples.size() / 2];
    if (median <= 0.f) median = 1e-6f;
    std::memcpy(&bits, &median, 4);
    cached.store(bits, std::memory_order_release);
  }
  std::memcpy(milliseconds, &bits, 4);
  return HY_OK;
}

hy_status hy_profile_read_kernel(uint32_t kernel, float* total_milliseconds, uint32_t* launches) {
  if (!total_milliseconds || !launches) return fail(HY_ERR_INVALID, "hy_profile_read_kernel: null argument");
  if (kernel >= HY_KERNEL_KINDS) return fail(HY_ERR_INVALID, "hy_profile_read_kernel: unknown kernel kind %u", kernel);
  return profile_sum(kernel, total_milliseconds, launches);
}

// Releases what the CALLING thread holds between calls: its scratch arena, its pool of temporary blocks, its pinned
// staging area, its join mailbox and its profiling events (all of them are thread-local and grow on demand, so calling the
// library again afterwards is fine).  Worker threads that end should call it: HIP objects freed from a thread_local destructor at
// process exit may outlive the runtime.
hy_status hy_shutdown(void) {
  (void)hipStreamSynchronize(t_stream);
  Scratch& s = scratch();
  if (s.base) (void)hipFree(s.base);
  if (s.ticket) (void)hipFree(s.ticket);
  if (s.status) (void)hipFree(s.status);
  s = Scratch{};
  for (auto& block : t_pool.free_blocks) (void)hipFree(block.second);
  t_pool.free_blocks.clear();
  if (t_staging.host) (void)hipHostFree(t_staging.host);
  t_staging.host = t_staging.device = nullptr;
  t_staging.bytes = 0;
  t_descriptor_ring.release();
  release_thread_join_state();
  for (hipEvent_t event : t_profile.events) (void)hipEventDestroy(event);
  t_profile.events.clear();
  t_profile.used = 0;
  t_profile.open = false;
  return HY_OK;
}

// The thread's temporaries (scratch arena, pooled blocks) are reused from call to call in stream order, so work still
// queued on the old stream has to finish before launches on a different stream may touch them.
hy_status hy_set_stream(void* hip_stream) {
  const auto stream = static_cast<hipStream_t>(hip_stream);
  if (stream != t_stream) (void)hipStreamSynchronize(t_stream);   // (the old handle may already be destroyed: then nothing is queued)
  t_stream = stream;
  return HY_OK;
}

hy_status hy_synchronize(void) {
  HY_HIP(hipStreamSynchronize(t_stream));
  return HY_OK;
}

hy_status hy_device_malloc(void** ptr, size_t bytes) {
  if (!ptr) return fail(HY_ERR_INVALID, "hy_device_malloc: null argument");
  bind_thread_device();
  HY_HIP(hipMalloc(ptr, bytes ? bytes : 256));
  return HY_OK;
}

hy_status hy_device_free(void* ptr) {
  if (ptr) HY_HIP(hipFree(ptr));
  return HY_OK;
}

hy_status hy_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  hipStream_t stream = current_stream();
  if (bytes) HY_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
  HY_HIP(hipStreamSynchronize(stream));
  return HY_OK;
}

hy_status hy_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  hipStream_t stream = current_stream();
  if (bytes) HY_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  return HY_OK;
}

// ---- residency cache -------------------------------------------------------------------------------------------------

static size_t type_width(uint32_t data_type) {
  switch (data_type) {
    case HY_TYPE_INT: return 4;
    case HY_TYPE_LONG: return 8;
    case HY_TYPE_FLOAT: return 4;
    case HY_TYPE_DOUBLE: return 8;
    default: return 0;
  }
}

constexpr uint32_t LZ4_MAX_BLOCK = 65536;   // decompressed bytes of an LZ4 block: its output is staged in LDS (Hyrise's blocks are 16 KB)

static hy_status validate_segment(const hy_segment& s, uint32_t chunk, uint32_t column_type) {
  if (s.data_type != column_type) return fail(HY_ERR_INVALID, "chunk %u: data type %u differs from chunk 0's %u", chunk, s.data_type, column_type);
  if (s.data_type < HY_TYPE_INT || s.data_type > HY_TYPE_STRING) return fail(HY_ERR_INVALID, "chunk %u: bad data type %u", chunk, s.data_type);
  switch (s.encoding) {
    case HY_ENC_UNENCODED:
      if (s.data_type == HY_TYPE_STRING) return fail(HY_ERR_UNSUPPORTED, "chunk %u: unencoded string segments stay on the CPU path", chunk);
      if (s.width != type_width(s.data_type)) return fail(HY_ERR_INVALID, "chunk %u: value width %u does not match type %u", chunk, s.width, s.data_type);
      if (s.size && !s.data) return fail(HY_ERR_INVALID, "chunk %u: null data pointer", chunk);
      break;
    case HY_ENC_DICTIONARY:
      if (s.width == 0 && (s.bits < 1 || s.bits > 32)) return fail(HY_ERR_INVALID, "chunk %u: bit-packed attribute vector of %u bits per element", chunk, s.bits);
      if (s.width != 0 && s.width != 1 && s.width != 2 && s.width != 4) return fail(HY_ERR_INVALID, "chunk %u: attribute vector width %u", chunk, s.width);
      if (s.size && !s.data) return fail(HY_ERR_INVALID, "chunk %u: null attribute vector", chunk);
      if (s.data_type != HY_TYPE_STRING && s.aux_size && !s.aux) return fail(HY_ERR_INVALID, "chunk %u: dictionary missing", chunk);
      break;
    case HY_ENC_FRAME_OF_REFERENCE:
      if (s.data_type != HY_TYPE_INT) return fail(HY_ERR_INVALID, "chunk %u: FrameOfReference is int32 only", chunk);
      if (s.width == 0 && (s.bits < 1 || s.bits > 32)) return fail(HY_ERR_INVALID, "chunk %u: bit-packed offset vector of %u bits per element", chunk, s.bits);
      if (s.width != 0 && s.width != 1 && s.width != 2 && s.width != 4) return fail(HY_ERR_INVALID, "chunk %u: offset width %u", chunk, s.width);
      if (s.size && (!s.data || !s.aux)) return fail(HY_ERR_INVALID, "chunk %u: FrameOfReference buffers missing", chunk);
      if (s.aux_size != (s.size + HY_FOR_BLOCK_SIZE - 1) / HY_FOR_BLOCK_SIZE) return fail(HY_ERR_INVALID, "chunk %u: %u block minima for %u rows", chunk, s.aux_size, s.size);
      break;
    case HY_ENC_RUN_LENGTH:
      if (s.data_type == HY_TYPE_STRING) return fail(HY_ERR_UNSUPPORTED, "chunk %u: run-length encoded strings stay on the CPU path", chunk);
      if (s.width != type_width(s.data_type)) return fail(HY_ERR_INVALID, "chunk %u: run value width %u does not match type %u", chunk, s.width, s.data_type);
      if (s.size && (!s.data || !s.aux || !s.aux_size)) return fail(HY_ERR_INVALID, "chunk %u: run-length buffers missing", chunk);
      break;
    case HY_ENC_MVCC:
      if (s.width != 4 || s.data_type != HY_TYPE_INT) return fail(HY_ERR_INVALID, "chunk %u: MVCC segments are three uint32 arrays (width 4, HY_TYPE_INT)", chunk);
      if (s.size && (!s.data || !s.aux || !s.nulls)) return fail(HY_ERR_INVALID, "chunk %u: MVCC arrays missing (tids, begin cids, end cids)", chunk);
      break;
    case HY_ENC_REFERENCE:
      if (!s.ref) return fail(HY_ERR_INVALID, "chunk %u: reference segment without referenced column", chunk);
      if (s.ref->is_reference) return fail(HY_ERR_INVALID, "chunk %u: reference segments must not reference reference segments (table_scan.cpp:140-148)", chunk);
      if (!s.data && s.ref_chunk_id >= s.ref->n_chunks) return fail(HY_ERR_INVALID, "chunk %u: EntireChunkPosList chunk id out of range", chunk);
      if (s.data && s.ref_chunk_id != 0xFFFFFFFFu && s.ref_chunk_id >= s.ref->n_chunks) return fail(HY_ERR_INVALID, "chunk %u: common chunk id out of range", chunk);
      break;
    case HY_ENC_LZ4: {
      if (s.data_type == HY_TYPE_STRING) return fail(HY_ERR_UNSUPPORTED, "chunk %u: LZ4 string segments stay on the CPU path", chunk);
      if (s.width != type_width(s.data_type)) return fail(HY_ERR_INVALID, "chunk %u: value width %u does not match type %u", chunk, s.width, s.data_type);
      const auto* lz4 = static_cast<const hy_lz4_blocks*>(s.data);
      if (!lz4) return fail(HY_ERR_INVALID, "chunk %u: LZ4 segment without its block descriptor", chunk);
      if (lz4->block_count && (!lz4->blocks || !lz4->block_bytes)) return fail(HY_ERR_INVALID, "chunk %u: LZ4 blocks missing", chunk);
      if (lz4->block_size > LZ4_MAX_BLOCK || lz4->last_block_size > lz4->block_size) return fail(HY_ERR_UNSUPPORTED, "chunk %u: LZ4 blocks of %u bytes (at most %u)", chunk, lz4->block_size, LZ4_MAX_BLOCK);
      const uint64_t decoded = lz4->block_count ? uint64_t{lz4->block_count - 1} * lz4->block_size + lz4->last_block_size : 0;
      if (decoded != uint64_t{s.size} * s.width) return fail(HY_ERR_INVALID, "chunk %u: LZ4 blocks decode to %llu bytes, %u rows of %u bytes expected", chunk, static_cast<unsigned long long>(decoded), s.size, s.width);
      if (lz4->dictionary_bytes && !lz4->dictionary) return fail(HY_ERR_INVALID, "chunk %u: LZ4 dictionary missing", chunk);
      break;
    }
    default: return fail(HY_ERR_UNSUPPORTED, "chunk %u: encoding %u stays on the CPU path", chunk, s.encoding);
  }
  if (s.sorted_by > HY_SORT_DESCENDING_NULLS_LAST) return fail(HY_ERR_INVALID, "chunk %u: sort mode %u", chunk, s.sorted_by);
  return HY_OK;
}

static bool bit_packed(const hy_segment& s) { return (s.encoding == HY_ENC_DICTIONARY || s.encoding == HY_ENC_FRAME_OF_REFERENCE) && s.width == 0; }
}  // extern "C"

namespace hy {

// ---- LZ4 blocks, decompressed on the device (lz4_Block_format.md; LZ4_decompress_safe_usingDict, lz4_segment.cpp:191-226) ---------------
// A block is a chain of sequences -- token (literal length << 4 | match length - 4, both continued in bytes of 255), the literals, a 2-byte
// match offset -- each of which copies bytes that are already there: one wavefront per block walks the chain with uniform (scalar) reads
// and copies with all 64 lanes, the block's output in LDS (an LDS instruction of a wave sees what its earlier ones wrote; global memory
// promises that only behind a wait for every store), which leaves as one coalesced copy.  A match may reach back in front of the block,
// into the segment's dictionary, and may overlap its own output (offset < length: the last `offset` bytes repeat).
struct Lz4Block {
  uint64_t source, target, dictionary;   // byte offsets: the compressed block and the dictionary inside `compressed`, the output inside `decoded`
  uint32_t source_bytes, target_bytes, dictionary_bytes, reserved;
};

__global__ __launch_bounds__(64) void lz4_decode_blocks(const uint8_t* compressed, uint8_t* decoded, const Lz4Block* blocks, uint32_t* error) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_out[];   // [target_bytes, rounded up to 16] the output | [source_bytes] the compressed block
  const uint32_t lane = threadIdx.x;
  const Lz4Block block = blocks[blockIdx.x];
  const uint8_t* dictionary = compressed + block.dictionary;
  const uint32_t n = block.source_bytes, size = block.target_bytes, history = block.dictionary_bytes;
  // the compressed block into LDS first (one coalesced read): the chain below reads it byte by byte, each read waiting for the one before
  uint8_t* in = s_out + ((size + 15) & ~15u);
  for (uint32_t k = lane; k < n; k += 64) in[k] = compressed[block.source + k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint32_t i = 0, o = 0;
  bool corrupt = false;
  while (i < n && !corrupt) {
    const uint32_t token = in[i++];
    uint32_t literals = token >> 4;
    if (literals == 15) {
      uint32_t extra;
      do {
        if (i >= n) { corrupt = true; break; }
        extra = in[i++];
        literals += extra;
      } while (extra == 255);
    }
    if (corrupt || i + literals > n || o + literals > size) { corrupt = true; break; }
    for (uint32_t k = lane; k < literals; k += 64) s_out[o + k] = in[i + k];
    i += literals;
    o += literals;
    if (i >= n) break;   // the last sequence ends after its literals
    if (i + 2 > n) { corrupt = true; break; }
    const uint32_t offset = in[i] | static_cast<uint32_t>(in[i + 1]) << 8;
    i += 2;
    uint32_t length = (token & 15) + 4;
    if ((token & 15) == 15) {
      uint32_t extra;
      do {
        if (i >= n) { corrupt = true; break; }
        extra = in[i++];
        length += extra;
      } while (extra == 255);
    }
    if (corrupt || offset == 0 || offset > o + history || o + length > size) { corrupt = true; break; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // byte k of the match is byte (k mod offset) of the `offset` bytes in front of the output position -- which may begin in the dictionary
    for (uint32_t k = lane; k < length; k += 64) {
      const int64_t from = static_cast<int64_t>(o) - offset + (k % offset);
      s_out[o + k] = from >= 0 ? s_out[from] : dictionary[history + from];
    }
    o += length;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (corrupt || o != size) { if (lane == 0) atomicOr(error, 1u); return; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint8_t* out = decoded + block.target;
  if (size % 4 == 0) { for (uint32_t k = lane; k < size / 4; k += 64) reinterpret_cast<uint32_t*>(out)[k] = reinterpret_cast<const uint32_t*>(s_out)[k]; }
  else { for (uint32_t k = lane; k < size; k += 64) out[k] = s_out[k]; }
}

// The LZ4 segments of a column that is being created: their blocks go to the device as they are, lz4_decode_blocks writes the values into a
// buffer the column owns, and the segments become the ValueSegments they were compressed from (data = device memory: on_device[c] = 1).
static hy_status decode_lz4_segments(hy_column* column, std::vector<uint8_t>& on_device) {
  std::vector<Lz4Block> blocks;
  std::vector<uint8_t> staged;
  size_t decoded_bytes = 0;
  std::vector<size_t> segment_target(column->n_chunks, 0);
  for (uint32_t c = 0; c < column->n_chunks; ++c) {
    const hy_segment& s = column->host_segments[c];
    if (s.encoding != HY_ENC_LZ4) continue;
    const auto* lz4 = static_cast<const hy_lz4_blocks*>(s.data);
    const size_t dictionary_at = staged.size();
    if (lz4->dictionary_bytes) staged.insert(staged.end(), static_cast<const uint8_t*>(lz4->dictionary), static_cast<const uint8_t*>(lz4->dictionary) + lz4->dictionary_bytes);
    segment_target[c] = decoded_bytes;
    for (uint32_t b = 0; b < lz4->block_count; ++b) {
      Lz4Block block{};
      block.source = staged.size();
      block.source_bytes = lz4->block_bytes[b];
      block.dictionary = dictionary_at;
      block.dictionary_bytes = lz4->dictionary_bytes;
      block.target = decoded_bytes + size_t{b} * lz4->block_size;
      block.target_bytes = b + 1 < lz4->block_count ? lz4->block_size : lz4->last_block_size;
      staged.insert(staged.end(), static_cast<const uint8_t*>(lz4->blocks[b]), static_cast<const uint8_t*>(lz4->blocks[b]) + lz4->block_bytes[b]);
      blocks.push_back(block);
    }
    decoded_bytes += align_up(size_t{s.size} * s.width + 16, 256);   // (+ 16: vector loads of the last partial group stay inside)
  }
  if (decoded_bytes == 0) return HY_OK;
  hipStream_t stream = current_stream();
  char* decoded = nullptr;
  HY_HIP(hipMalloc(reinterpret_cast<void**>(&decoded), decoded_bytes));
  column->owned.push_back(decoded);
  if (!blocks.empty()) {
    DeviceBuffer d_staged, d_blocks, d_error;
    HY_TRY(d_staged.alloc(staged.size() + 16));
    HY_TRY(d_blocks.alloc(sizeof(Lz4Block) * blocks.size()));
    HY_TRY(d_error.alloc(4));
    HY_HIP(hipMemcpyAsync(d_staged.ptr, staged.data(), staged.size(), hipMemcpyHostToDevice, stream));
    HY_HIP(hipMemcpyAsync(d_blocks.ptr, blocks.data(), sizeof(Lz4Block) * blocks.size(), hipMemcpyHostToDevice, stream));
    HY_HIP(hipMemsetAsync(d_error.ptr, 0, 4, stream));
    static OncePerDevice lds_raised;
    uint64_t device_bit = 0;
    if (lds_raised.pending(&device_bit)) {
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_decode_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LZ4_MAX_BLOCK + 4096));
      lds_raised.done(device_bit);
    }
    uint32_t largest = 0;   // LDS per block: its output and its compressed bytes
    for (const Lz4Block& block : blocks) largest = std::max(largest, ((block.target_bytes + 15) & ~15u) + block.source_bytes);
    if (largest > 2 * LZ4_MAX_BLOCK + 4096) return fail(HY_ERR_UNSUPPORTED, "an LZ4 block of %u bytes (compressed + decompressed) does not fit LDS", largest);
    hipLaunchKernelGGL(lz4_decode_blocks, dim3(static_cast<uint32_t>(blocks.size())), dim3(64), (largest + 15) & ~15u, stream, d_staged.as<uint8_t>(),
                       reinterpret_cast<uint8_t*>(decoded), d_blocks.as<Lz4Block>(), d_error.as<uint32_t>());
    uint32_t error = 0;
    HY_HIP(hipMemcpyAsync(&error, d_error.ptr, 4, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));   // (the staged blocks and the host vectors above live until here)
    if (error) return fail(HY_ERR_INVALID, "an LZ4 block is corrupt (its sequences do not decode to the block's size)");
  }
  for (uint32_t c = 0; c < column->n_chunks; ++c) {
    hy_segment& s = column->host_segments[c];
    if (s.encoding != HY_ENC_LZ4) continue;
    s.encoding = HY_ENC_UNENCODED;
    s.data = decoded + segment_target[c];
    s.aux = nullptr;
    s.aux_size = 0;
    on_device[c] = 1;
  }
  return HY_OK;
}

}  // namespace hy

extern "C" {

static bool compressed(const hy_segment& s) { return s.encoding == HY_ENC_RUN_LENGTH || bit_packed(s); }

static size_t data_bytes(const hy_segment& s) {
  if (s.encoding == HY_ENC_RUN_LENGTH) return size_t{s.width} * s.aux_size;                 // one value per run
  if (bit_packed(s)) return 8 * ((size_t{s.size} * s.bits + 63) / 64);                       // compact::vector: whole 64-bit words
  return s.encoding == HY_ENC_REFERENCE ? (s.data ? size_t{8} * s.size : 0) : size_t{s.width} * s.size;
}
static size_t aux_bytes(const hy_segment& s) {
  if (s.encoding == HY_ENC_RUN_LENGTH) return size_t{4} * s.aux_size;                        // inclusive end positions
  if (s.encoding == HY_ENC_DICTIONARY) return s.aux ? type_width(s.data_type) * s.aux_size : 0;
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) return size_t{4} * s.aux_size;
  if (s.encoding == HY_ENC_MVCC) return size_t{4} * s.size;
  return 0;
}
static size_t null_bytes(const hy_segment& s) {
  if (s.encoding == HY_ENC_MVCC) return size_t{4} * s.size;   // the end commit ids travel in the `nulls` slot
  if (s.encoding == HY_ENC_RUN_LENGTH) return s.nulls ? size_t{s.aux_size} : 0;   // one byte per run
  return (s.nulls && s.encoding != HY_ENC_REFERENCE) ? size_t{8} * ((s.size + 63) / 64) : 0;
}

hy_status hy_column_create(const hy_segment* segments, uint32_t n_chunks, uint32_t mem, hy_column** out) {
  if (!out) return fail(HY_ERR_INVALID, "hy_column_create: null output");
  bind_thread_device();
  *out = nullptr;
  if (n_chunks && !segments) return fail(HY_ERR_INVALID, "hy_column_create: null segments");
  if (mem != HY_MEM_HOST && mem != HY_MEM_DEVICE) return fail(HY_ERR_INVALID, "hy_column_create: bad memory space %u", mem);
  const uint32_t column_type = n_chunks ? segments[0].data_type : HY_TYPE_INT;
  for (uint32_t c = 0; c < n_chunks; ++c) HY_TRY(validate_segment(segments[c], c, column_type));

  auto column = new hy_column();
  column->n_chunks = n_chunks;
  column->device = t_bound_device >= 0 ? t_bound_device : 0;
  column->data_type = column_type;
  column->host_segments.assign(segments, segments + n_chunks);
  column->row_base.resize(size_t{n_chunks} + 1, 0);
  auto cleanup = [&](hy_status st) {
    hy_column_destroy(column);
    return st;
  };

  // RunLengthSegments and bit-packed vectors travel and stay as they are (TableScan reads them in place; plain_column decodes a twin
  // on the device for the operators that gather rows).  Their runs are checked here, on the host, while they still are host memory.
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const hy_segment& s = segments[c];
    if (compressed(s)) column->has_compressed = true;
    if (s.sorted_by != HY_SORT_NONE) column->has_sorted = true;
    if (s.encoding != HY_ENC_RUN_LENGTH || mem != HY_MEM_HOST) continue;
    const auto* run_ends = static_cast<const uint32_t*>(s.aux);
    uint32_t row = 0;
    for (uint32_t run = 0; run < s.aux_size; ++run) {
      if (run_ends[run] >= s.size || run_ends[run] < row) return cleanup(fail(HY_ERR_INVALID, "chunk %u: run %u ends at %u (rows: %u)", c, run, run_ends[run], s.size));
      row = run_ends[run] + 1;
    }
    if (row != s.size) return cleanup(fail(HY_ERR_INVALID, "chunk %u: runs cover %u of %u rows", c, row, s.size));
    // A segment without a NULL run travels without its flags: IS NULL / IS NOT NULL then take the early-outs of a ValueSegment without a
    // null vector (column_is_null_table_scan_impl.cpp:197-251) -- what the decoded twin of the segment is.
    bool any_null = false;
    for (uint32_t run = 0; run < s.aux_size && s.nulls; ++run) any_null = any_null || reinterpret_cast<const uint8_t*>(s.nulls)[run] != 0;
    if (!any_null) column->host_segments[c].nulls = nullptr;
  }
  // LZ4 segments: decompressed on the device, ValueSegments from here on (their values are device memory already)
  std::vector<uint8_t> on_device(n_chunks, 0);
  bool any_lz4 = false;
  for (uint32_t c = 0; c < n_chunks; ++c) any_lz4 = any_lz4 || segments[c].encoding == HY_ENC_LZ4;
  if (any_lz4) {
    if (mem != HY_MEM_HOST) return cleanup(fail(HY_ERR_INVALID, "LZ4 segments are handed over as host memory (HY_MEM_HOST)"));
    const hy_status decoded = decode_lz4_segments(column, on_device);
    if (decoded != HY_OK) return cleanup(decoded);
  }
  segments = column->host_segments.data();

  // One arena for every buffer of the column: 916 chunks x 3 buffers would otherwise be ~2.7k hipMallocs.
  size_t arena_bytes = 0;
  if (mem == HY_MEM_HOST) {
    for (uint32_t c = 0; c < n_chunks; ++c) {
      const hy_segment& s = segments[c];
      // +16: vector loads of the last partial group never leave the allocation
      arena_bytes += (on_device[c] ? 0 : align_up(data_bytes(s) + 16, 256)) + align_up(aux_bytes(s) + 16, 256) + align_up(null_bytes(s) + 16, 256);
    }
  }
  char* arena = nullptr;
  if (arena_bytes) {
    hipError_t err = hipMalloc(reinterpret_cast<void**>(&arena), arena_bytes);
    if (err != hipSuccess) return cleanup(fail(HY_ERR_DEVICE, "hipMalloc(%zu) failed: %s", arena_bytes, hipGetErrorString(err)));
    column->owned.push_back(arena);
  }
  // The buffers travel through pinned memory, 32 MiB at a time: they are copied side by side -- in the layout of the arena -- into one half of
  // this thread's pinned block while the other half is on its way, and leave as ONE transfer per window.  (A hipMemcpy per buffer from the
  // caller's pageable memory -- 2 748 of them for a 916-chunk dictionary column -- moved l_shipdate at 1 GB/s; the link does 55.)
  constexpr size_t UPLOAD_WINDOW = size_t{32} << 20;
  size_t cursor = 0, window_begin = 0;
  uint32_t half = 0;
  char* pinned = nullptr;
  hipEvent_t window_sent[2] = {nullptr, nullptr};
  bool window_busy[2] = {false, false};
  struct WindowEvents {   // (destroyed on every way out)
    hipEvent_t* events;
    ~WindowEvents() { for (int i = 0; i < 2; ++i) if (events[i]) (void)hipEventDestroy(events[i]); }
  } window_events{window_sent};
  // The copies into the window are collected and made by several threads when the window is sent: one thread moves pageable memory into
  // the pinned block at 16 - 20 GB/s, a third of what the link takes.
  struct WindowCopy { char* to; const void* from; size_t bytes; };
  std::vector<WindowCopy> window_copies;
  auto copy_window = [&]() {
    size_t total = 0;
    for (const WindowCopy& c : window_copies) total += c.bytes;
    const unsigned helpers = total >= (size_t{4} << 20) ? std::min(4u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    auto work = [&](unsigned t) {   // contiguous shares of about total / helpers bytes
      const size_t share_begin = total * t / helpers, share_end = total * (t + 1) / helpers;
      size_t at = 0;
      for (const WindowCopy& c : window_copies) {
        const size_t begin = std::max(at, share_begin), end = std::min(at + c.bytes, share_end);
        if (begin < end) std::memcpy(c.to + (begin - at), static_cast<const char*>(c.from) + (begin - at), end - begin);
        at += c.bytes;
        if (at >= share_end) break;
      }
    };
    if (helpers == 1) work(0);
    else {
      std::vector<std::thread> threads;
      for (unsigned t = 1; t < helpers; ++t) threads.emplace_back(work, t);
      work(0);
      for (std::thread& thread : threads) thread.join();
    }
    window_copies.clear();
  };
  auto flush = [&]() -> hipError_t {
    if (cursor == window_begin) return hipSuccess;
    copy_window();
    hipError_t err = hipMemcpyAsync(arena + window_begin, pinned + half * UPLOAD_WINDOW, cursor - window_begin, hipMemcpyHostToDevice, t_stream);
    if (err == hipSuccess) err = hipEventRecord(window_sent[half], t_stream);
    window_busy[half] = true;
    half ^= 1;
    window_begin = cursor;
    if (err == hipSuccess && window_busy[half]) { err = hipEventSynchronize(window_sent[half]); window_busy[half] = false; }
    return err;
  };
  auto upload = [&](const void* src, size_t bytes, const void** dst) -> hipError_t {
    *dst = nullptr;
    if (!src || !bytes) return hipSuccess;
    if (!pinned) {
      void* host = nullptr;
      void* device = nullptr;
      if (pinned_staging(2 * UPLOAD_WINDOW, &host, &device) != HY_OK) return hipErrorOutOfMemory;
      pinned = static_cast<char*>(host);
      for (hipEvent_t& e : window_sent) { const hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (err != hipSuccess) return err; }
    }
    const size_t padded = align_up(bytes + 16, 256);
    hipError_t err = hipSuccess;
    if (cursor + padded - window_begin > UPLOAD_WINDOW) err = flush();
    char* target = arena + cursor;
    *dst = target;
    if (err == hipSuccess && padded > UPLOAD_WINDOW) {   // (larger than a window: straight from the caller's memory)
      err = hipMemcpyAsync(target, src, bytes, hipMemcpyHostToDevice, t_stream);
      cursor += padded;
      window_begin = cursor;
      return err;
    }
    if (err == hipSuccess) window_copies.push_back(WindowCopy{pinned + half * UPLOAD_WINDOW + (cursor - window_begin), src, bytes});
    cursor += padded;
    return err;
  };

  std::vector<DevSegment> dev(n_chunks ? n_chunks : 1);
  std::vector<Slice> slices;
  std::vector<Part> parts;
  uint32_t part_slices = PART_SLICES;   // slices per part: one Hyrise chunk (65 535 rows) = one part = one workgroup
  if (FIXED_PART_SLICES > 0) part_slices = static_cast<uint32_t>(FIXED_PART_SLICES);
  if (part_slices < 1) part_slices = 1;
  if (part_slices > PART_SLICES) part_slices = PART_SLICES;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hy_segment& s = column->host_segments[c];
    DevSegment& d = dev[c];
    std::memset(&d, 0, sizeof(d));
    if (mem == HY_MEM_HOST) {
      const void *p_data = on_device[c] ? s.data : nullptr, *p_aux = nullptr, *p_nulls = nullptr;
      hipError_t err = on_device[c] ? hipSuccess : upload(s.data, data_bytes(s), &p_data);
      if (err == hipSuccess) err = upload(s.aux, aux_bytes(s), &p_aux);
      if (err == hipSuccess) err = upload(s.nulls, null_bytes(s), &p_nulls);
      if (err != hipSuccess) return cleanup(fail(HY_ERR_DEVICE, "chunk %u upload failed: %s", c, hipGetErrorString(err)));
      s.data = p_data;
      s.aux = p_aux;
      s.nulls = static_cast<const uint64_t*>(p_nulls);
    }
    d.data = s.data;
    d.aux = s.aux;
    d.nulls = s.encoding == HY_ENC_REFERENCE ? nullptr : s.nulls;
    d.size = s.size;
    d.aux_size = s.aux_size;
    d.ref_chunk_id = s.ref_chunk_id;
    d.encoding = static_cast<uint8_t>(s.encoding);
    d.data_type = static_cast<uint8_t>(s.data_type);
    d.width = static_cast<uint8_t>(bit_packed(s) ? SEG_PACKED | s.bits : s.width);
    d.flags = static_cast<uint8_t>(s.sorted_by << SEG_SORT_SHIFT);
    if (reinterpret_cast<uintptr_t>(s.data) % 16 != 0 || reinterpret_cast<uintptr_t>(s.nulls) % 8 != 0) d.flags |= SEG_UNALIGNED;
    if (s.encoding == HY_ENC_REFERENCE) {
      column->is_reference = true;
      if (column->ref && column->ref != s.ref) return cleanup(fail(HY_ERR_INVALID, "chunk %u references a different table than chunk 0", c));
      column->ref = s.ref;
      const hy_column* referenced = nullptr;   // (a referenced column with run-length / bit-packed segments is read through its decoded twin)
      const hy_status plain_status = plain_column(s.ref, &referenced);
      if (plain_status != HY_OK) return cleanup(plain_status);
      d.ref = referenced->d_segments;
      if (s.data && s.ref_chunk_id == 0xFFFFFFFFu) column->multi_chunk_reference = true;
    } else if (column->is_reference) {
      return cleanup(fail(HY_ERR_INVALID, "chunk %u: data and reference segments mixed in one column", c));
    }
    if (s.encoding == HY_ENC_DICTIONARY && !s.aux && s.aux_size) column->has_dictionary_without_values = true;
    if (s.encoding == HY_ENC_MVCC) column->is_mvcc = true;
    else if (column->is_mvcc) return cleanup(fail(HY_ERR_INVALID, "chunk %u: MVCC and data segments mixed in one column", c));
    column->row_base[c + 1] = column->row_base[c] + s.size;
    uint32_t begin = 0;
    const uint32_t first_slice_of_chunk = static_cast<uint32_t>(slices.size());
    do {
      const uint32_t count = (s.size - begin < SLICE_ROWS) ? s.size - begin : SLICE_ROWS;
      slices.push_back(Slice{c, begin, count, begin == 0 ? 1u : 0u});
      begin += count;
    } while (begin < s.size);
    const uint32_t chunk_slices = static_cast<uint32_t>(slices.size()) - first_slice_of_chunk;
    const uint32_t chunk_parts = (chunk_slices + part_slices - 1) / part_slices;
    const uint32_t first_part = static_cast<uint32_t>(parts.size());
    for (uint32_t p = 0; p < chunk_parts; ++p) {
      const uint32_t first = first_slice_of_chunk + p * part_slices;
      const uint32_t n = (chunk_slices - p * part_slices < part_slices) ? chunk_slices - p * part_slices : part_slices;
      parts.push_back(Part{first, n, c, p, chunk_parts, first_part, p * part_slices * SLICE_ROWS, 0, column->row_base[c]});
    }
  }
  // Streaming instantiation of the scan kernel: one common element width, aligned buffers, 32-bit comparisons.
  {   // the last window; the host waits for the stream below (descriptor upload), before anybody else writes into the pinned block
    const hipError_t err = flush();
    if (err != hipSuccess) return cleanup(fail(HY_ERR_DEVICE, "upload failed: %s", hipGetErrorString(err)));
  }
  uint32_t stream_width = 0;
  bool streamable = n_chunks > 0 && !column->is_reference;
  for (uint32_t c = 0; c < n_chunks && streamable; ++c) {
    const hy_segment& s = column->host_segments[c];
    const bool kind_ok = s.encoding == HY_ENC_DICTIONARY || s.encoding == HY_ENC_FRAME_OF_REFERENCE ||
                         (s.encoding == HY_ENC_UNENCODED && s.data_type == HY_TYPE_INT);
    // (width class 16: a BitPackingVector of at most 16 bits per element -- the streaming kernel unpacks eight elements from 20 bytes)
    const uint32_t width_class = bit_packed(s) ? (s.bits <= 16 ? 16u : 0u) : s.width;
    if (!kind_ok || (dev[c].flags & SEG_UNALIGNED) || s.encoding == HY_ENC_RUN_LENGTH || width_class == 0) streamable = false;
    else if (stream_width == 0) stream_width = width_class;
    else if (stream_width != width_class) streamable = false;
  }
  column->stream_width = streamable ? stream_width : 0;
  column->rows = column->row_base[n_chunks];
  column->n_slices = static_cast<uint32_t>(slices.size());
  column->n_parts = static_cast<uint32_t>(parts.size());

  std::vector<SliceView> views(slices.size());
  for (size_t i = 0; i < slices.size(); ++i) {
    const hy_segment& s = column->host_segments[slices[i].chunk];
    SliceView& v = views[i];
    v.data = s.data;
    v.aux = s.aux;
    v.chunk = slices[i].chunk;
    v.row_begin = slices[i].row_begin;
    v.row_count = slices[i].row_count;
    v.kind = VIEW_GENERIC;
    const bool aligned = reinterpret_cast<uintptr_t>(s.data) % 16 == 0;   // the kernels that use views read 16 bytes per lane
    if (aligned && s.encoding == HY_ENC_UNENCODED && s.data_type == HY_TYPE_INT && !s.nulls) v.kind = VIEW_INT32;
    if (aligned && s.encoding == HY_ENC_FRAME_OF_REFERENCE && !s.nulls && s.width != 0) v.kind = s.width == 1 ? VIEW_FOR8 : s.width == 2 ? VIEW_FOR16 : VIEW_FOR32;
  }
  // The five descriptor tables travel as ONE block with ONE copy (operator chains create a reference column per operator and
  // column: five allocations and five synchronous copies each were most of what a chain's step cost on the host).
  auto aligned = [](size_t bytes) { return (bytes + 255) & ~size_t{255}; };
  const size_t segments_bytes = sizeof(DevSegment) * dev.size(), slices_bytes = sizeof(Slice) * slices.size(), views_bytes = sizeof(SliceView) * views.size();
  const size_t row_base_bytes = 8 * (size_t{n_chunks} + 1), parts_bytes = sizeof(Part) * parts.size();
  const size_t at_slices = aligned(segments_bytes + sizeof(DevSegment)), at_views = at_slices + aligned(slices_bytes + sizeof(Slice));
  const size_t at_row_base = at_views + aligned(views_bytes + sizeof(SliceView)), at_parts = at_row_base + aligned(row_base_bytes);
  const size_t at_first_slice = at_parts + aligned(parts_bytes + sizeof(Part));
  const size_t total = at_first_slice + aligned(4 * (size_t{n_chunks} + 1));
  std::vector<uint32_t> first_slice(size_t{n_chunks} + 1, 0);
  for (size_t i = slices.size(); i-- > 0;) first_slice[slices[i].chunk] = static_cast<uint32_t>(i);   // (descending: the chunk's first slice is written last)
  first_slice[n_chunks] = static_cast<uint32_t>(slices.size());
  // where the tables are put together: a slot of the pinned ring (columns over device memory whose tables fit one), else a vector
  std::vector<unsigned char> staging_vector;
  unsigned char* staging = nullptr;
  int ring_slot = -1;
  if (mem == HY_MEM_DEVICE && total <= DescriptorRing::SLOT_BYTES) {
    DescriptorRing& ring = t_descriptor_ring;
    if (!ring.host && hipHostMalloc(reinterpret_cast<void**>(&ring.host), DescriptorRing::SLOTS * DescriptorRing::SLOT_BYTES, hipHostMallocDefault) != hipSuccess) ring.host = nullptr;
    if (ring.host) {
      ring_slot = static_cast<int>(ring.next);
      ring.next = (ring.next + 1) % DescriptorRing::SLOTS;
      if (!ring.sent[ring_slot] && hipEventCreateWithFlags(&ring.sent[ring_slot], hipEventDisableTiming) != hipSuccess) ring_slot = -1;
      else if (ring.busy[ring_slot]) { (void)hipEventSynchronize(ring.sent[ring_slot]); ring.busy[ring_slot] = false; }
    }
    if (ring_slot >= 0) {
      staging = ring.host + size_t{static_cast<uint32_t>(ring_slot)} * DescriptorRing::SLOT_BYTES;
      std::memset(staging, 0, total);
    }
  }
  if (!staging) {
    staging_vector.assign(total, 0);
    staging = staging_vector.data();
  }
  if (segments_bytes) std::memcpy(staging, dev.data(), segments_bytes);
  if (slices_bytes) std::memcpy(staging + at_slices, slices.data(), slices_bytes);
  if (views_bytes) std::memcpy(staging + at_views, views.data(), views_bytes);
  std::memcpy(staging + at_row_base, column->row_base.data(), row_base_bytes);
  if (parts_bytes) std::memcpy(staging + at_parts, parts.data(), parts_bytes);
  std::memcpy(staging + at_first_slice, first_slice.data(), 4 * (size_t{n_chunks} + 1));
  void* block = nullptr;
  size_t block_capacity = 0;
  if (pool_acquire(total, &block, &block_capacity) != HY_OK) return cleanup(HY_ERR_DEVICE);
  column->pooled.emplace_back(block_capacity, block);
  column->descriptors_pooled = true;
  unsigned char* base = static_cast<unsigned char*>(block);
  column->d_segments = reinterpret_cast<DevSegment*>(base);
  column->d_slices = reinterpret_cast<Slice*>(base + at_slices);
  column->d_slice_views = reinterpret_cast<SliceView*>(base + at_views);
  column->d_row_base = reinterpret_cast<uint64_t*>(base + at_row_base);
  column->d_parts = reinterpret_cast<Part*>(base + at_parts);
  column->d_first_slice = reinterpret_cast<uint32_t*>(base + at_first_slice);
  // The block comes from this thread's pool: whoever held it before may have released it while kernels on the thread's stream still
  // read it (pool blocks are recycled in STREAM order).  The upload therefore goes onto that stream -- a copy on the NULL stream is not
  // ordered behind a non-blocking stream -- and the host waits for it (`staging` dies with this call).
  // (From the ring nobody waits: the slot outlives the copy.  From the vector -- and for every column that was uploaded from host memory
  // above -- the host waits: `staging_vector` dies with this call, and the caller's buffers are his again when it returns.)
  hipError_t err = hipMemcpyAsync(block, staging, total, hipMemcpyHostToDevice, t_stream);
  if (err == hipSuccess && ring_slot >= 0) {
    err = hipEventRecord(t_descriptor_ring.sent[ring_slot], t_stream);
    t_descriptor_ring.busy[ring_slot] = err == hipSuccess;
  } else if (err == hipSuccess) err = hipStreamSynchronize(t_stream);
  if (err != hipSuccess) return cleanup(fail(HY_ERR_DEVICE, "descriptor upload failed: %s", hipGetErrorString(err)));
  *out = column;
  return HY_OK;
}

}  // extern "C"

namespace hy {

// ---- decoded twins of run-length / bit-packed segments (hy_column::plain) -------------------------------------------------------------
// The compressed buffers are resident; these kernels write what RunLengthSegmentIterable (run_length_segment_iterable.hpp:100-160: row r
// belongs to the first run whose inclusive end position is >= r) and the BitPacking decompressor (bitpacking_decompressor.hpp:35-37:
// element i = bits [i * b, (i + 1) * b) of the little-endian word stream) hand out, once, into buffers the twin column owns.
struct ExpandTarget {
  void* values;        // RunLength: T values[size]; bit-packed: FixedWidthInteger vector of `width` bytes per element
  uint64_t* nulls;     // RunLength with NULL runs: the ValueSegment's null bitmap, else nullptr
  uint32_t width;
};

__global__ __launch_bounds__(256) void expand_compressed(const DevSegment* segments, const ExpandTarget* targets, uint32_t n_chunks) {
  const uint32_t chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const DevSegment s = segments[chunk];
  const ExpandTarget t = targets[chunk];
  if (!t.values) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  if (s.encoding == HY_ENC_RUN_LENGTH) {
    const uint32_t* ends = static_cast<const uint32_t*>(s.aux);
    const uint8_t* run_is_null = reinterpret_cast<const uint8_t*>(s.nulls);
    for (uint32_t base = 0; base < s.size; base += 256) {   // (whole waves: the ballot below needs every lane)
      const uint32_t row = base + tid;
      bool is_null = false;
      if (row < s.size) {
        uint32_t low = 0, high = s.aux_size - 1;   // first run with ends[run] >= row
        while (low < high) {
          const uint32_t middle = (low + high) / 2;
          if (ends[middle] >= row) high = middle; else low = middle + 1;
        }
        is_null = run_is_null && run_is_null[low] != 0;
        if (s.width == 8) static_cast<uint64_t*>(t.values)[row] = is_null ? 0ull : static_cast<const uint64_t*>(s.data)[low];
        else static_cast<uint32_t*>(t.values)[row] = is_null ? 0u : static_cast<const uint32_t*>(s.data)[low];
      }
      const uint64_t word = __ballot(is_null);
      if (t.nulls && lane == 0 && base + (tid & ~63u) < s.size) t.nulls[(base + tid) >> 6] = word;
    }
    return;
  }
  const uint64_t* words = static_cast<const uint64_t*>(s.data);
  const uint32_t bits = seg_bits(s);
  const uint64_t mask = (1ull << bits) - 1;
  for (uint32_t row = tid; row < s.size; row += 256) {
    const uint64_t at = uint64_t{row} * bits;
    const uint32_t shift = static_cast<uint32_t>(at & 63);
    uint64_t value = words[at >> 6] >> shift;
    if (shift + bits > 64) value |= words[(at >> 6) + 1] << (64 - shift);
    value &= mask;
    if (t.width == 1) static_cast<uint8_t*>(t.values)[row] = static_cast<uint8_t>(value);
    else if (t.width == 2) static_cast<uint16_t*>(t.values)[row] = static_cast<uint16_t>(value);
    else static_cast<uint32_t*>(t.values)[row] = static_cast<uint32_t>(value);
  }
}

hy_status plain_column(const hy_column* column, const hy_column** plain) {
  *plain = column;
  if (!column || !column->has_compressed) return HY_OK;
  std::lock_guard<std::mutex> lock(column->plain_mutex);
  if (column->plain) { *plain = column->plain; return HY_OK; }
  bind_thread_device();
  const uint32_t n_chunks = column->n_chunks;
  std::vector<hy_segment> segments(column->host_segments);
  std::vector<ExpandTarget> targets(n_chunks, ExpandTarget{nullptr, nullptr, 0});
  size_t arena_bytes = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const hy_segment& s = segments[c];
    if (s.encoding == HY_ENC_RUN_LENGTH) arena_bytes += align_up(size_t{s.width} * s.size + 16, 256) + (s.nulls ? align_up(8 * ((size_t{s.size} + 63) / 64) + 16, 256) : 0);
    else if (compressed(s)) arena_bytes += align_up(size_t{s.bits <= 8 ? 1u : s.bits <= 16 ? 2u : 4u} * s.size + 16, 256);
  }
  char* arena = nullptr;
  DeviceBuffer d_targets;
  HY_TRY(d_targets.alloc(sizeof(ExpandTarget) * n_chunks));
  HY_HIP(hipMalloc(reinterpret_cast<void**>(&arena), arena_bytes ? arena_bytes : 256));
  size_t cursor = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hy_segment& s = segments[c];
    if (!compressed(s)) continue;
    ExpandTarget& t = targets[c];
    if (s.encoding == HY_ENC_RUN_LENGTH) {
      t.width = s.width;
      t.values = arena + cursor;
      cursor += align_up(size_t{s.width} * s.size + 16, 256);
      if (s.nulls) { t.nulls = reinterpret_cast<uint64_t*>(arena + cursor); cursor += align_up(8 * ((size_t{s.size} + 63) / 64) + 16, 256); }
      s.encoding = HY_ENC_UNENCODED;
      s.data = t.values;
      s.aux = nullptr;
      s.aux_size = 0;
      s.nulls = t.nulls;
    } else {
      t.width = s.bits <= 8 ? 1u : s.bits <= 16 ? 2u : 4u;
      t.values = arena + cursor;
      cursor += align_up(size_t{t.width} * s.size + 16, 256);
      s.width = t.width;
      s.bits = 0;
      s.data = t.values;
    }
  }
  hipStream_t stream = current_stream();
  hipError_t err = hipMemcpyAsync(d_targets.ptr, targets.data(), sizeof(ExpandTarget) * n_chunks, hipMemcpyHostToDevice, stream);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(expand_compressed, dim3(n_chunks), dim3(256), 0, stream, column->d_segments, d_targets.as<ExpandTarget>(), n_chunks);
    err = hipStreamSynchronize(stream);   // (`targets` dies with this call; other threads may read the twin as soon as it is published)
  }
  hy_column* twin = nullptr;
  hy_status status = err == hipSuccess ? hy_column_create(segments.data(), n_chunks, HY_MEM_DEVICE, &twin) : fail(HY_ERR_DEVICE, "decoding a compressed column failed: %s", hipGetErrorString(err));
  if (status != HY_OK) { (void)hipFree(arena); return status; }
  twin->owned.push_back(arena);
  column->plain = twin;
  *plain = twin;
  return HY_OK;
}

}  // namespace hy

extern "C" {

hy_status hy_column_destroy(hy_column* column) {
  if (!column) return HY_OK;
  // The descriptor block goes back to this thread's pool and is handed out again in the order of this thread's stream: wait for that
  // stream only (a column that other threads still use must not be destroyed: the caller's contract, as for any shared object) --
  // a device-wide synchronise here stalled every thread's stream at every step of an operator chain.
  // A column of ANOTHER device (one process, one worker thread per GPU: multi_gpu.hpp's shards die on the coordinator's thread): its
  // pooled blocks must not enter this thread's pool -- the pool hands blocks out for kernels of this thread's device -- and the stream to
  // wait for is not this thread's: free them (hipFree waits for the device's work).
  bind_thread_device();
  const bool foreign = t_bound_device >= 0 && column->device != t_bound_device;
  if (column->descriptors_pooled && !foreign) (void)hipStreamSynchronize(t_stream);
  if (column->plain) (void)hy_column_destroy(column->plain);
  for (void* p : column->owned) (void)hipFree(p);
  for (auto& block : column->pooled) {
    if (foreign) (void)hipFree(block.second);
    else pool_release(block.second, block.first);
  }
  if (!column->descriptors_pooled) {
    if (column->d_segments) (void)hipFree(column->d_segments);
    if (column->d_slices) (void)hipFree(column->d_slices);
    if (column->d_slice_views) (void)hipFree(column->d_slice_views);
    if (column->d_parts) (void)hipFree(column->d_parts);
    if (column->d_row_base) (void)hipFree(column->d_row_base);
  }
  delete column;
  return HY_OK;
}

hy_status hy_column_row_count(const hy_column* column, uint64_t* rows) {
  if (!column || !rows) return fail(HY_ERR_INVALID, "hy_column_row_count: null argument");
  *rows = column->rows;
  return HY_OK;
}

hy_status hy_column_chunk_count(const hy_column* column, uint32_t* chunks) {
  if (!column || !chunks) return fail(HY_ERR_INVALID, "hy_column_chunk_count: null argument");
  *chunks = column->n_chunks;
  return HY_OK;
}

}  // extern "C"
