// hy_device.hpp -- internal declarations shared by the HIP translation units behind include/hyrise_amd.h.
// gfx950 (MI355X / CDNA4) only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hyrise_amd.h"
#include "hy_options.hpp"

namespace hy {

// ---- device-side view of one segment (48 bytes, read through the scalar/L2 path by every workgroup) ---------------
struct DevSegment {
  const void* data;            // values | attribute vector | FoR offsets | pos list (hy_row_id)
  const void* aux;             // dictionary | block minima
  const uint64_t* nulls;       // null bitmap or nullptr
  const DevSegment* ref;       // REFERENCE: segments of the referenced column
  uint32_t size;
  uint32_t aux_size;
  uint32_t ref_chunk_id;
  uint8_t encoding;
  uint8_t data_type;
  uint8_t width;               // bytes per element of `data`; SEG_PACKED | b: a bit-packed attribute / offset vector of b bits per element
  uint8_t flags;               // SEG_UNALIGNED | HY_SORT_* << SEG_SORT_SHIFT
};
enum : uint8_t { SEG_UNALIGNED = 1, SEG_SORT_SHIFT = 4, SEG_PACKED = 0x80 };
// (the struct stays 48 bytes -- three 16-byte scalar loads, and the streaming scan keeps two of them in scalar registers)
__host__ __device__ inline bool seg_is_packed(const DevSegment& s) { return (s.width & SEG_PACKED) != 0; }
__host__ __device__ inline uint32_t seg_bits(const DevSegment& s) { return s.width & 0x7Fu; }
__host__ __device__ inline uint32_t seg_sorted_by(const DevSegment& s) { return s.flags >> SEG_SORT_SHIFT; }

// A contiguous run of at most SLICE_ROWS rows of one chunk: the unit of work of the scan / materialise kernels.
struct Slice {
  uint32_t chunk;
  uint32_t row_begin;
  uint32_t row_count;
  uint32_t first_of_chunk;     // 1 if row_begin == 0
};

// What a kernel that runs AHEAD of its data needs to know about a slice, in one 32-byte record (one scalar load instead of
// the slice -> segment descriptor -> data chain): where the rows' stored words are and how to read them.
struct SliceView {
  const void* data;            // the chunk's values / offsets (VIEW_GENERIC: unused, the kernel goes through DevSegment)
  const void* aux;             // FrameOfReference: block minima
  uint32_t chunk;
  uint32_t row_begin;
  uint32_t row_count;
  uint32_t kind;               // VIEW_*
};
enum : uint32_t { VIEW_GENERIC = 0, VIEW_INT32 = 1, VIEW_FOR8 = 2, VIEW_FOR16 = 3, VIEW_FOR32 = 4 };   // int32 values | FoR offsets of 1 / 2 / 4 bytes, no NULLs

// A part: at most PART_SLICES consecutive slices of ONE chunk -- the unit a scan workgroup owns.  A Hyrise chunk
// (<= 65 535 rows) is exactly one part, so its PosList is produced by one workgroup without any inter-workgroup traffic.
struct Part {
  uint32_t first_slice;
  uint32_t n_slices;
  uint32_t chunk;
  uint32_t part_in_chunk;
  uint32_t parts_in_chunk;
  uint32_t first_part;          // index of the chunk's part 0
  uint32_t first_row;           // chunk offset of the part's first row
  uint32_t reserved;
  uint64_t region_base;         // first RowID slot of the chunk's output region (= rows in all chunks before it)
};
constexpr uint32_t PART_SLICES = 8;   // upper bound (LDS sizing); the actual part length is chosen per column

constexpr uint32_t WG_THREADS = 256;
constexpr uint32_t ROWS_PER_THREAD = 32;
constexpr uint32_t SLICE_ROWS = WG_THREADS * ROWS_PER_THREAD;   // 8192

// hipFuncSetAttribute applies to the calling thread's current device: a kernel that asks for more dynamic LDS than a workgroup gets by default
// is registered once per DEVICE and call site (a process-wide flag would leave the second GPU of a one-process plan -- hy_bind_device, one
// worker thread per GPU -- with the default limit and its launches failing).
class OncePerDevice {
 public:
  bool pending(uint64_t* bit) const {
    int device = 0;
    (void)hipGetDevice(&device);
    *bit = 1ull << (static_cast<unsigned>(device) & 63u);
    return (done_.load(std::memory_order_acquire) & *bit) == 0;
  }
  void done(uint64_t bit) { done_.fetch_or(bit, std::memory_order_release); }

 private:
  std::atomic<uint64_t> done_{0};
};

}  // namespace hy

// What the first JoinHash over a resident build column learned about its keys (join.hip: rank_table_fill_checked): later joins
// fill their rank table in one pass sized by it and CHECK it in the same pass -- it saves no work on the keys, only a host round
// trip and the second read.  state: 0 nothing known, 1 valid, 2 do not use (a check failed once).
struct hy_join_key_hint {
  std::atomic<uint32_t> state{0};
  std::atomic<uint32_t> unique{0};          // no key twice (a column with duplicates still serves Semi / Anti joins: presence bits only)
  std::atomic<uint64_t> key_min{0}, key_max{0};
  std::atomic<uint32_t> has_duplicates{0};  // rank_table_mark met a key twice: later joins over the column go straight to the sorted directory
  std::atomic<uint32_t> probe_locality{0};  // as a PROBE side: 0 not looked at, 1 neighbouring rows hold neighbouring keys, 2 they do not (probe_key_locality, join.hip)
};

// The per-chunk jobs of a ColumnVsValue / Between / IsNull predicate (prepare_jobs, scan.hip: one dictionary bound search and the all /
// none early-outs per chunk) are a pure function of the column's segments -- immutable once encoded, abstract_encoded_segment.hpp:12-17 --
// and of the predicate's literal words: a data column keeps the job tables of the last few predicates it was scanned with (a prepared
// statement's scan, a benchmark's repeated query), later scans with the same literal read them instead of searching 916 dictionaries
// again (one launch, 6 us at SF10 in front of a 60 us scan).  Every row is still tested by the scan kernel.
struct hy_scan_job_cache {
  struct Entry {
    uint8_t key[40];        // condition | value_type | value | value2 | column_is_nullable | materialize_all | no_ranges
    void* jobs = nullptr;   // [n_chunks + 1] ScanJob, device memory owned by the column
    hipStream_t stream = nullptr;   // the stream the jobs were prepared on: scans on another stream prepare their own
    uint64_t used = 0;
  };
  std::mutex mutex;
  Entry entries[4];
  uint64_t clock = 0;
};

// The opaque ABI type.
struct hy_column {
  uint32_t n_chunks = 0;
  uint32_t data_type = HY_TYPE_NULL;
  uint64_t rows = 0;
  int device = 0;                           // the device of the thread that created it: operators refuse a column of another device
  bool is_reference = false;
  bool is_mvcc = false;                     // HY_ENC_MVCC segments: a table's MvccData, only hy_validate reads it
  bool multi_chunk_reference = false;       // some pos list spans several referenced chunks
  bool has_dictionary_without_values = false;
  uint32_t stream_width = 0;                // 1|2|4 if every segment is an aligned W-byte id/offset/int32 vector, else 0
  const hy_column* ref = nullptr;
  std::vector<hy_segment> host_segments;    // caller's descriptors (pointers patched to device addresses)
  std::vector<uint64_t> row_base;           // [n_chunks + 1] prefix sum of sizes
  hy::DevSegment* d_segments = nullptr;
  hy::Slice* d_slices = nullptr;
  hy::SliceView* d_slice_views = nullptr;   // [n_slices]
  uint32_t n_slices = 0;
  hy::Part* d_parts = nullptr;
  uint32_t n_parts = 0;
  uint64_t* d_row_base = nullptr;           // [n_chunks + 1] device copy of row_base
  uint32_t* d_first_slice = nullptr;        // [n_chunks + 1] index of every chunk's first slice (the last entry: n_slices)
  bool descriptors_pooled = false;          // the five descriptor tables above are pieces of ONE pooled block (column->pooled), uploaded with one copy
  std::vector<void*> owned;                 // device allocations freed with the column
  std::vector<std::pair<size_t, void*>> pooled;   // ... or handed back to the buffer pool (operator results)
  mutable hy_join_key_hint join_hint;
  mutable hy_scan_job_cache scan_jobs;
  // smallest / largest non-NULL value of an int32 data column (join_star.hpp: the range a dimension's direct table spans): computed once --
  // the segments are immutable -- by the first star join that uses the column as a dimension key.  state 1: valid, 2: the column holds no value
  mutable std::atomic<uint32_t> extent_state{0};
  mutable std::atomic<int64_t> extent_min{0}, extent_max{0};
  mutable std::atomic<uint64_t> aggregate_hint{0};   // aggregate.hip: which path the last GROUP BY led by this column ended on (signature of the column set << 8 | partition bits + 1)
  mutable std::atomic<uint64_t> star_aggregate_hint{0};   // plan.hip: the same for the GROUP BY of a star join whose first GROUP BY column this is (its aggregate reads intermediate columns)
  // RunLength segments and bit-packed vectors stay compressed in device memory; TableScan reads them in place.  The operators that
  // gather rows (joins, aggregates, projections, exchanges, reference columns) read `plain`: the same column as Value / FixedWidthInteger
  // segments, decoded ON THE DEVICE from the resident compressed buffers the first time one of them asks (plain_column, runtime.hip).
  bool has_compressed = false;
  bool has_sorted = false;                  // some chunk is flagged as sorted by this column (hy_segment::sorted_by): its scan may take JOB_RANGE jobs
  mutable std::mutex plain_mutex;
  mutable hy_column* plain = nullptr;       // owned
};

namespace hy {

// ---- errors ---------------------------------------------------------------------------------------------------------
hy_status fail(hy_status code, const char* fmt, ...);
hy_status on_this_device(const hy_column* column, const char* entry_point);   // runtime.hip: the calling thread's device holds the column
hy_status plain_column(const hy_column* column, const hy_column** plain);      // runtime.hip: `column` itself, or its decoded twin (see hy_column::plain)
#define HY_HIP(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t err__ = (expr);                                                                            \
    if (err__ != hipSuccess)                                                                              \
      return ::hy::fail(HY_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
  } while (0)
#define HY_TRY(expr)                    \
  do {                                  \
    hy_status st__ = (expr);            \
    if (st__ != HY_OK) return st__;     \
  } while (0)

hipStream_t current_stream();
void bind_thread_device();   // hipSetDevice(the device hy_init chose) once per thread

// Pinned, device-mapped host memory of this thread (grow-only): small results that kernels store straight into host
// memory, read by the host after a stream synchronise -- instead of one blit kernel and host round trip per hipMemcpyAsync.
hy_status pinned_staging(size_t bytes, void** host, void** device);

// join.hip: stable LSD radix sort of (key, id) pairs by key (see there)
hy_status sort_pairs_u32(uint32_t** keys, uint32_t** ids, uint32_t* keys_tmp, uint32_t* ids_tmp, uint64_t n, uint32_t key_bits, hipStream_t stream);

// Optional HIP-event bracket around the dominant kernel of an operator call (hy_set_profiling / hy_last_kernel_ms).
void release_thread_join_state();   // join.hip: frees the calling thread's pinned mailbox (hy_shutdown)
void profile_begin(hipStream_t stream, uint32_t kind = HY_KERNEL_OTHER);   // kind: HY_KERNEL_* (hy_profile_read_kernel)
void profile_end(hipStream_t stream);
bool profile_events(hipEvent_t* start, hipEvent_t* stop, uint32_t kind = HY_KERNEL_OTHER);   // per-kernel pair for hipExtLaunchKernelGGL

// ---- per-thread scratch (status words for decoupled look-back, job tables, temporaries) ----------------------------
// Every operator call of a thread reuses one growing arena; thread-safe because it is thread-local, as are streams.
struct Scratch {
  void* base = nullptr;
  size_t capacity = 0;
  size_t used = 0;
  uint32_t* ticket = nullptr;     // monotonically increasing work-ticket counter (never reset: base passed per launch)
  uint32_t ticket_base = 0;
  uint32_t epoch = 0;             // tag of the current launch in look-back status words (30 bits, never 0)
  uint64_t* status = nullptr;     // look-back status words; used for NOTHING else, so stale words carry older epochs
  size_t status_capacity = 0;
  hy_status reserve(size_t bytes);   // ensure capacity (may reallocate: only call before carving)
  hy_status begin_launch(size_t status_words, uint32_t tickets);  // new epoch; returns with status/ticket_base valid
  void* carve(size_t bytes);         // 256-byte aligned sub-allocation of the arena
  void reset() { used = 0; }
};
Scratch& scratch();

template <typename T>
inline T* carve(Scratch& s, size_t count) {
  return static_cast<T*>(s.carve(sizeof(T) * (count ? count : 1)));
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


// Temporary device buffers come from a per-thread pool of power-of-two blocks that is reused across calls in stream
// order (hipMalloc/hipFree cost ~100 us each and synchronise the device); hy_shutdown releases the calling thread's pool.
hy_status pool_acquire(size_t bytes, void** ptr, size_t* capacity);
void pool_release(void* ptr, size_t capacity);

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t capacity = 0;
  bool borrowed = false;   // `ptr` belongs to somebody else (borrow): nothing is given back
  hy_status alloc(size_t bytes);
  void borrow(void* block);
  ~DeviceBuffer();
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  template <typename T> T* as() const { return static_cast<T*>(ptr); }
};


// ---- aggregate.hip: which path a GROUP BY should take, for callers whose input columns do not live long enough to remember it themselves
// (hy_column::aggregate_hint does that for resident columns; the intermediate tables of hy_star_join_aggregate are new in every call).
// t_aggregate_recommended: after a hy_aggregate_hash of this thread -- 0: aggregate_rows is fine, b + 1: start on the partitioned path with b
// bits next time (the call ended there, or it met more groups than a workgroup's LDS table holds in an input of a few million rows).
// t_aggregate_next_path: set before a call -- b + 1 starts that call on the partitioned path with b bits (consumed by the call).
extern thread_local uint32_t t_aggregate_recommended, t_aggregate_next_path;

// exchange.hip: hy_column_export with the chunks' first rows given (device array [n_chunks], in elements; nullptr: the column's own)
hy_status export_column_at(const hy_column* column, void* values, uint8_t* nulls, const uint64_t* d_row_base);

// ---- join_star.hpp (join.hip): the probes of a star join fused into one pass over the fact table (hy_star_join_aggregate, plan.hip) ------
struct StarProbeDimension {
  const hy_column* key;        // the dimension's key column (int32, unique among `rows`)
  const hy_row_id* rows;       // the dimension rows that take part (device memory): the rows that pass its filter, or all of them -- or nullptr:
                               // every row of the key column's table that passes `filter` (tested by the kernels that build the dimension's tables)
  const hy_column* filter;     // rows == nullptr: the column the filter tests (a data column of the dimension's table), or nullptr: no filter
  const struct ScanJob* filter_jobs;   // ... and its per-chunk jobs in device memory (prepare_scan_jobs)
  uint64_t n_rows;             // how many -- or, with d_n_rows / a filter, at most how many
  const uint64_t* d_n_rows;    // the count in device memory (a scan's total that no host has read), or nullptr
  const hy_column* fact_key;   // the fact table's foreign key to this dimension
  bool want_rows;              // the caller reads columns of this dimension at the surviving rows
};
// fact_rows / dimension_rows[d]: the RowIDs of the fact table and of dimension d per row of fact JOIN dim_1 ... JOIN dim_k (Inner), in the
// fact table's row order.  *applicable = false: this shape is not for the fused probe (keys that are not int32 / not unique / too
// sparse, foreign-key segments the probe does not stream) -- nothing was produced.
// The plan's Projection + AggregateHash inside the join (star_finish, join_star.hpp): asked for with `finish`, answered in `groups` --
// groups->done: the join result was grouped where it was found (no RowIDs were written); otherwise the RowIDs are there as without it.
struct StarFinishColumnSpec { uint32_t table; const hy_column* column; };   // table 0 = the fact table, d + 1 = dimension d; column nullptr: none
struct StarFinishRequest {
  uint32_t n_groupby, n_aggregates;
  StarFinishColumnSpec groupby[4];
  struct { uint32_t function, op; StarFinishColumnSpec left, right; } aggregates[8];
};
struct StarFinishGroups {
  bool done = false;
  uint32_t n_groups = 0;
  uint32_t input_type[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // type of every aggregate's input value (HY_TYPE_INT / HY_TYPE_LONG)
  std::vector<uint64_t> keys, first, last, values, counts;   // [n][4], [n], [n], [n][8], [n]: first / last = row numbers of the join result
};
hy_status star_probe_rows(const StarProbeDimension* dimensions, uint32_t n_dimensions, DeviceBuffer& fact_rows, std::vector<std::unique_ptr<DeviceBuffer>>& dimension_rows,
                          uint64_t* n_rows, bool* applicable, const StarFinishRequest* finish = nullptr, StarFinishGroups* groups = nullptr);
hy_status star_all_rows_of(const hy_column* column, DeviceBuffer& rows);   // every row of a data column's table as a PosList
// scan.hip: the kernels of hy_poslist_translate queued on the current stream, nothing read back -- dense_offsets: [n_chunks + 1] device words,
// the last of them the number of RowIDs written
hy_status poslist_translate_queued(const hy_column* scanned, const hy_scan_result* result, uint32_t layout, hy_row_id* out, uint64_t capacity, uint64_t* dense_offsets);

}  // namespace hy
