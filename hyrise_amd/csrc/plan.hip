// plan.hip -- a star join as ONE call over the operator entry points (BASELINE.json configs[4]: the SSB star joins).
//
// Hyrise runs such a query as an operator tree: a TableScan per filtered dimension, one JoinHash per dimension with the filtered dimension as
// build side and the join result so far as probe side, a Projection for the aggregates' expressions, an AggregateHash.  Between the
// operators the reference hands reference tables on (PosLists); here the intermediates stay in HBM, and this function is the adapter's
// plan for that shape: nothing but calls of hy_table_scan / hy_poslist_translate / hy_column_create / hy_column_export / hy_join_hash /
// hy_gather_row_ids / hy_projection_arithmetic / hy_aggregate_hash, in the order hyrise_amd/ssb.py run_query makes them -- without an
// interpreter, a tensor allocation and a ctypes marshalling between two of them (a quarter of an SSB query was idle device between
// operator calls).  Intermediates are presented to the next operator as tables of DENSE_CHUNK rows:
//   * a filtered dimension's keys, the foreign keys of the join result and the columns the aggregate reads are MATERIALISED
//     (hy_column_export through the PosLists: JoinHash materialises its inputs anyway, join_hash_steps.hpp:274-330) into plain value
//     columns, which the primary-key / foreign-key join kernels and the aggregate's streaming decoders read with wide loads;
//   * per joined table the base RowIDs of every surviving row are carried along (hy_gather_row_ids with the probe positions).
#include "hy_device.hpp"
#include "hy_scan_job.hpp"

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

namespace hy {

static thread_local int t_last_star_was_fused = 0;   // debug / tests: the thread's last hy_star_join_aggregate took the fused probe (2: and grouped inside it)

constexpr uint32_t DENSE_CHUNK = 65536;   // rows per chunk of an intermediate table (chunks start on 16-byte boundaries for every type; smaller chunks -- more,
                                          // smaller slices for the aggregate -- were tried: 2048-row chunks cost more on the host, 700 descriptors per column, than they gave)

struct ColumnHandle;
// Columns a step of the plan is done with: hy_column_destroy waits for the thread's stream (their descriptor blocks go back to the pool),
// which -- right behind the launch that reads them -- kept the host from queueing the next step until the device had caught up (a third
// of an SSB query was such waits).  They are kept until the plan's last kernel is queued and go together.
static thread_local std::vector<std::unique_ptr<ColumnHandle>>* t_plan_done_with = nullptr;

struct ColumnHandle {   // an hy_column this plan created
  hy_column* column = nullptr;
  ColumnHandle() = default;
  ColumnHandle(const ColumnHandle&) = delete;
  ColumnHandle& operator=(const ColumnHandle&) = delete;
  ~ColumnHandle() { reset(); }
  void reset() { if (column) (void)hy_column_destroy(column); column = nullptr; }
};

static size_t type_bytes(uint32_t data_type) { return (data_type == HY_TYPE_INT || data_type == HY_TYPE_FLOAT) ? 4 : 8; }

// `rows` RowIDs into the data column `base`, presented as ReferenceSegments of DENSE_CHUNK rows (read in place)
static hy_status reference_column(const hy_column* base, const hy_row_id* rows, uint64_t n, ColumnHandle& out, uint32_t chunk_rows = DENSE_CHUNK) {
  const uint32_t n_chunks = static_cast<uint32_t>(std::max<uint64_t>(1, (n + chunk_rows - 1) / chunk_rows));
  std::vector<hy_segment> segments(n_chunks);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hy_segment& s = segments[c];
    std::memset(&s, 0, sizeof(s));
    const uint64_t begin = uint64_t{c} * chunk_rows, end = std::min<uint64_t>(n, begin + chunk_rows);
    s.encoding = HY_ENC_REFERENCE;
    s.data_type = base->data_type;
    s.size = static_cast<uint32_t>(end > begin ? end - begin : 0);
    s.width = 8;
    s.data = rows + begin;
    s.ref = base;
    s.ref_chunk_id = 0xFFFFFFFFu;
  }
  return hy_column_create(segments.data(), n_chunks, HY_MEM_DEVICE, &out.column);
}

// `values` (n elements of `data_type`, device memory) as ValueSegments of DENSE_CHUNK rows; null_words: one bit per row (bit r & 63 of word
// r >> 6: a ValueSegment's null vector, chunk after chunk -- chunk_rows is a multiple of 64), or nullptr: no NULLs
static hy_status value_column(const void* values, uint64_t n, uint32_t data_type, ColumnHandle& out, uint32_t chunk_rows = DENSE_CHUNK, const uint64_t* null_words = nullptr) {
  const uint32_t n_chunks = static_cast<uint32_t>(std::max<uint64_t>(1, (n + chunk_rows - 1) / chunk_rows));
  const size_t width = type_bytes(data_type);
  std::vector<hy_segment> segments(n_chunks);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hy_segment& s = segments[c];
    std::memset(&s, 0, sizeof(s));
    const uint64_t begin = uint64_t{c} * chunk_rows, end = std::min<uint64_t>(n, begin + chunk_rows);
    s.encoding = HY_ENC_UNENCODED;
    s.data_type = data_type;
    s.size = static_cast<uint32_t>(end > begin ? end - begin : 0);
    s.width = static_cast<uint32_t>(width);
    s.data = static_cast<const char*>(values) + begin * width;
    s.nulls = null_words ? null_words + begin / 64 : nullptr;
    s.ref_chunk_id = 0xFFFFFFFFu;
  }
  return hy_column_create(segments.data(), n_chunks, HY_MEM_DEVICE, &out.column);
}

// Could a cell of the column be NULL as far as its layout tells?  (Value / FrameOfReference segments without a null vector cannot; a dictionary
// segment's NULL is a value id, which only the data shows.)
static bool may_hold_nulls(const hy_column* column) {
  for (const hy_segment& s : column->host_segments) {
    if (s.nulls != nullptr || (s.encoding != HY_ENC_UNENCODED && s.encoding != HY_ENC_FRAME_OF_REFERENCE)) return true;
  }
  return false;
}

__global__ __launch_bounds__(256) void any_null_byte(const uint8_t* bytes, uint64_t n, uint32_t* found) {
  bool any = false;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) any = any || bytes[i] != 0;
  if (__any(any) && (threadIdx.x & 63) == 0) *found = 1;
}

// one bit per row out of one byte per row (hy_column_export's NULL flags -> a ValueSegment's null vector); n: a multiple of 64 rows are written
__global__ __launch_bounds__(256) void null_bytes_to_words(const uint8_t* bytes, uint64_t n, uint64_t* words) {
  const uint64_t padded = (n + 63) & ~uint64_t{63};
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < padded; i += static_cast<uint64_t>(gridDim.x) * 256) {
    const uint64_t lanes = __ballot(i < n && bytes[i] != 0);
    if ((threadIdx.x & 63) == 0) words[i >> 6] = lanes;
  }
}

// column `base` at the rows `rows` as a plain value column (values -- and, where a cell is NULL, the null vector -- in `storage`): a nullable
// foreign key whose NULL finds no partner, a GROUP BY column whose NULLs are a group of their own, an aggregate input whose NULLs are skipped
// (round 6; before, a NULL cell sent the whole plan back to the caller's operator chain)
static hy_status materialise(const hy_column* base, const hy_row_id* rows, uint64_t n, DeviceBuffer& storage, ColumnHandle& out, uint32_t chunk_rows = DENSE_CHUNK) {
  if (base->data_type < HY_TYPE_INT || base->data_type > HY_TYPE_DOUBLE) return fail(HY_ERR_UNSUPPORTED, "hy_star_join_aggregate: numeric columns only");
  const size_t values_bytes = (type_bytes(base->data_type) * std::max<uint64_t>(n, 1) + 16 + 255) & ~size_t{255};
  const size_t words_bytes = 8 * ((std::max<uint64_t>(n, 1) + 63) / 64) + 64;
  const bool nullable = n != 0 && may_hold_nulls(base);
  HY_TRY(storage.alloc(values_bytes + (nullable ? words_bytes : 0)));
  const uint64_t* null_words = nullptr;
  if (n) {
    auto through_owner = std::make_unique<ColumnHandle>();
    ColumnHandle& through = *through_owner;
    struct Later {   // (on every way out of this block)
      std::unique_ptr<ColumnHandle>& owner;
      ~Later() { if (t_plan_done_with) t_plan_done_with->push_back(std::move(owner)); }
    } later{through_owner};
    HY_TRY(reference_column(base, rows, n, through, chunk_rows));
    if (may_hold_nulls(base)) {
      DeviceBuffer null_bytes, found;
      HY_TRY(null_bytes.alloc(n));
      HY_TRY(found.alloc(64));
      hipStream_t stream = current_stream();
      HY_HIP(hipMemsetAsync(found.ptr, 0, 4, stream));
      HY_TRY(hy_column_export(through.column, storage.ptr, null_bytes.as<uint8_t>()));
      hipLaunchKernelGGL(any_null_byte, dim3(static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 2048))), dim3(256), 0, stream, null_bytes.as<uint8_t>(), n, found.as<uint32_t>());
      uint32_t any = 0;
      HY_HIP(hipMemcpyAsync(&any, found.ptr, 4, hipMemcpyDeviceToHost, stream));
      HY_HIP(hipStreamSynchronize(stream));
      if (any) {
        uint64_t* words = reinterpret_cast<uint64_t*>(storage.as<char>() + values_bytes);
        hipLaunchKernelGGL(null_bytes_to_words, dim3(static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 2048))), dim3(256), 0, stream, null_bytes.as<uint8_t>(), n, words);
        null_words = words;   // (null_bytes goes back to this thread's pool when the block ends: handed out again in stream order, behind this kernel)
      }
    } else {
      HY_TRY(hy_column_export(through.column, storage.ptr, nullptr));
    }
  }
  return value_column(storage.ptr, n, base->data_type, out, chunk_rows, null_words);
}

// The rows of `filter_column` that satisfy the predicate, as one dense PosList in device memory
static hy_status filtered_rows(const hy_column* filter_column, const hy_predicate* predicate, DeviceBuffer& rows, uint64_t* n_rows, DeviceBuffer* count_in_memory = nullptr) {
  DeviceBuffer regions, offsets, counts;
  const uint64_t capacity = std::max<uint64_t>(1, filter_column->rows);
  HY_TRY(regions.alloc(sizeof(hy_row_id) * capacity));
  HY_TRY(offsets.alloc(8 * (size_t{filter_column->n_chunks} + 1)));
  HY_TRY(counts.alloc(4 * std::max<size_t>(1, filter_column->n_chunks)));
  HY_TRY(rows.alloc(sizeof(hy_row_id) * capacity));
  hy_scan_result scan;
  std::memset(&scan, 0, sizeof(scan));
  scan.mem = HY_MEM_DEVICE;
  scan.flags = HY_SCAN_CHUNK_REGIONS | HY_SCAN_MATERIALIZE_ALL_MATCH;
  scan.matches = regions.as<hy_row_id>();
  scan.capacity = capacity;
  scan.offsets = offsets.as<uint64_t>();
  scan.counts = counts.as<uint32_t>();
  HY_TRY(hy_table_scan(filter_column, predicate, nullptr, 0, &scan));
  if (!count_in_memory) return hy_poslist_translate(filter_column, &scan, HY_POSLIST_DENSE, rows.as<hy_row_id>(), capacity, n_rows);
  // (the fused probe: the kernels that read the rows read their number from device memory too -- no host in between)
  HY_TRY(count_in_memory->alloc(8 * (size_t{filter_column->n_chunks} + 2)));
  *n_rows = capacity;
  return poslist_translate_queued(filter_column, &scan, HY_POSLIST_DENSE, rows.as<hy_row_id>(), capacity, count_in_memory->as<uint64_t>());
}

struct JoinOutput {
  DeviceBuffer arena, slice_offsets;
  hy_row_id* left = nullptr;
  hy_row_id* right = nullptr;
  uint64_t n_pairs = 0;
};

// build x probe (Inner), PosLists in device memory: both from one allocation, the second 1.25 MiB past a 2 MiB boundary
static hy_status join_inner(const hy_column* build, const hy_column* probe, JoinOutput& out) {
  constexpr size_t PERIOD = size_t{2} << 20, OFFSET = size_t{5} << 18;
  uint64_t capacity = std::max<uint64_t>({1, build->rows, probe->rows});
  uint64_t slice_capacity = std::max(build->rows, probe->rows) / 131070 + std::max(build->n_chunks, probe->n_chunks) + 600;
  for (int attempt = 0; attempt < 3; ++attempt) {
    HY_TRY(out.slice_offsets.alloc(8 * (slice_capacity + 2)));
    const size_t list_bytes = sizeof(hy_row_id) * capacity;
    HY_TRY(out.arena.alloc(2 * list_bytes + 3 * PERIOD));
    char* base = out.arena.as<char>();
    char* first = base + (PERIOD - reinterpret_cast<uintptr_t>(base) % PERIOD) % PERIOD;
    char* second = first + (list_bytes + PERIOD - 1) / PERIOD * PERIOD + OFFSET;
    out.left = reinterpret_cast<hy_row_id*>(first);
    out.right = reinterpret_cast<hy_row_id*>(second);
    hy_join_result r;
    std::memset(&r, 0, sizeof(r));
    r.mem = HY_MEM_DEVICE;
    r.radix_bits = 0xFFFFFFFFu;
    r.left_pos = out.left;
    r.right_pos = out.right;
    r.capacity = capacity;
    r.slice_offsets = out.slice_offsets.as<uint64_t>();
    r.slice_capacity = slice_capacity;
    const hy_status status = hy_join_hash(build, probe, HY_JOIN_INNER, &r);
    // (the result did not fit what was offered: the call says what it needs -- pairs, output PosLists, or both; like operators.join, once each)
    if (status == HY_ERR_CAPACITY && attempt < 2 && (r.n_pairs > capacity || r.n_slices > slice_capacity)) {
      capacity = std::max<uint64_t>(capacity, r.n_pairs);
      slice_capacity = std::max<uint64_t>(slice_capacity, r.n_slices);
      continue;
    }
    HY_TRY(status);
    out.n_pairs = r.n_pairs;
    return HY_OK;
  }
  return fail(HY_ERR_CAPACITY, "hy_star_join_aggregate: a join did not fit the capacity it asked for");
}

// The groups star_finish left (join_star.hpp) -> the aggregate's result, by AggregateHash's rules as hy_aggregate_hash applies them to the
// join result (aggregate.hip run_aggregate): rows ordered by the groups' first rows, or by ascending key under the immediate-key shortcut
// (one int GROUP BY column whose key range is below 1.2 x the rows, aggregate_hash.cpp:388-401, 770-804: the representative row is then the
// group's LAST row); representative rows as RowIDs of the intermediate table the chain would have aggregated (DENSE_CHUNK rows per chunk);
// result types of window_function_traits.hpp:11-77.  No cell is NULL (star_finish refuses NULL inputs).
static hy_status write_star_groups(const StarFinishGroups& groups, const hy_star_column* groupby, uint32_t n_groupby, const hy_star_aggregate* aggregates, uint32_t n_aggregates,
                                   uint64_t n_rows, hy_aggregate_result* result) {
  const uint32_t n = groups.n_groups;
  result->n_groups = n;
  if (n > result->group_capacity) return fail(HY_ERR_CAPACITY, "aggregate produces %u groups, capacity is %u", n, result->group_capacity);
  bool immediate = false;
  auto sort_key_of = [&](uint32_t i) { return static_cast<uint64_t>(static_cast<int64_t>(groups.keys[size_t{i} * 4]) - static_cast<int64_t>(INT32_MIN)) + 1; };
  if (n_groupby == 1 && groupby[0].column->data_type == HY_TYPE_INT && n) {
    uint64_t min_key = ~0ull, max_key = 0;
    for (uint32_t i = 0; i < n; ++i) { min_key = std::min(min_key, sort_key_of(i)); max_key = std::max(max_key, sort_key_of(i)); }
    immediate = max_key > 0 && static_cast<double>(max_key - min_key) < static_cast<double>(n_rows) * 1.2;
  }
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return immediate ? sort_key_of(x) < sort_key_of(y) : groups.first[x] < groups.first[y]; });
  for (uint32_t o = 0; o < n && result->group_row_ids; ++o) {
    const uint64_t row = immediate ? groups.last[order[o]] : groups.first[order[o]];
    result->group_row_ids[o] = hy_row_id{static_cast<uint32_t>(row / DENSE_CHUNK), static_cast<uint32_t>(row % DENSE_CHUNK)};
  }
  for (uint32_t g = 0; g < n_aggregates; ++g) {
    hy_aggregate_column& col = result->columns[g];
    const uint32_t function = aggregates[g].function, in_type = groups.input_type[g];
    col.data_type = function == HY_AGG_COUNT ? static_cast<uint32_t>(HY_TYPE_LONG) : function == HY_AGG_AVG ? static_cast<uint32_t>(HY_TYPE_DOUBLE) : function == HY_AGG_SUM ? static_cast<uint32_t>(HY_TYPE_LONG) : in_type;
    if (!col.values) return fail(HY_ERR_INVALID, "aggregate %u: values buffer missing", g);
    for (uint32_t o = 0; o < n; ++o) {
      const int64_t value = static_cast<int64_t>(groups.values[size_t{order[o]} * 8 + g]);
      const uint64_t count = groups.counts[order[o]];
      if (col.is_null) col.is_null[o] = 0;
      if (function == HY_AGG_COUNT) static_cast<int64_t*>(col.values)[o] = static_cast<int64_t>(count);
      else if (function == HY_AGG_AVG) static_cast<double*>(col.values)[o] = static_cast<double>(value) / static_cast<double>(count);   // (the chain adds doubles of integers: exact below 2^53)
      else if (col.data_type == HY_TYPE_INT) static_cast<int32_t*>(col.values)[o] = static_cast<int32_t>(value);
      else static_cast<int64_t*>(col.values)[o] = value;
    }
  }
  return HY_OK;
}

}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_star_join_aggregate(const hy_star_dimension* dimensions, uint32_t n_dimensions, const hy_star_column* groupby, uint32_t n_groupby,
                                 const hy_star_aggregate* aggregates, uint32_t n_aggregates, hy_aggregate_result* result, uint64_t* joined_rows) {
  if (!dimensions || !n_dimensions || n_dimensions > HY_MAX_STAR_DIMENSIONS) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: 1 .. %d dimensions", static_cast<int>(HY_MAX_STAR_DIMENSIONS));
  if ((n_groupby && !groupby) || (n_aggregates && !aggregates) || !result) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: null argument");
  if (n_aggregates > HY_MAX_STAR_AGGREGATES || n_groupby > HY_MAX_STAR_AGGREGATES) return fail(HY_ERR_UNSUPPORTED, "hy_star_join_aggregate: at most %d GROUP BY columns / aggregates", static_cast<int>(HY_MAX_STAR_AGGREGATES));
  auto table_ok = [&](const hy_star_column& c) { return c.column != nullptr && c.table <= n_dimensions; };
  for (uint32_t g = 0; g < n_groupby; ++g) if (!table_ok(groupby[g])) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: GROUP BY column %u names no table of the join", g);
  for (uint32_t a = 0; a < n_aggregates; ++a) {
    const hy_star_aggregate& spec = aggregates[a];
    if (spec.left.column && !table_ok(spec.left)) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: aggregate %u names no table of the join", a);
    if (spec.op != HY_STAR_NO_OP && (!spec.left.column || !table_ok(spec.right))) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: aggregate %u: an expression needs two columns", a);
  }
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    if (!dimensions[d].key || !dimensions[d].fact_key) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: dimension %u: key columns missing", d);
    HY_TRY(on_this_device(dimensions[d].key, "hy_star_join_aggregate"));
    HY_TRY(on_this_device(dimensions[d].fact_key, "hy_star_join_aggregate"));
    if (dimensions[d].filter_column) HY_TRY(on_this_device(dimensions[d].filter_column, "hy_star_join_aggregate"));
  }
  // RowIDs found in one column of a table index the others: the columns named for a table must agree in their chunks (the fact table: with
  // its first foreign key; dimension d: with its key) -- a mismatch would be read out of bounds by the kernels that dereference them
  auto same_table = [](const hy_column* x, const hy_column* y) {
    if (x->n_chunks != y->n_chunks || x->rows != y->rows) return false;
    for (uint32_t c = 0; c < x->n_chunks; ++c) if (x->host_segments[c].size != y->host_segments[c].size) return false;
    return true;
  };
  auto table_column = [&](uint32_t table) { return table == 0 ? dimensions[0].fact_key : dimensions[table - 1].key; };
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    if (!same_table(dimensions[d].fact_key, dimensions[0].fact_key)) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: dimension %u: its foreign key is not a column of the fact table the others name", d);
    if (dimensions[d].filter_column && !same_table(dimensions[d].filter_column, dimensions[d].key)) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: dimension %u: filter and key column are not columns of one table", d);
  }
  auto check_column = [&](const hy_star_column& c, const char* what, uint32_t index) -> hy_status {
    if (!c.column) return HY_OK;
    HY_TRY(on_this_device(c.column, "hy_star_join_aggregate"));
    if (!same_table(c.column, table_column(c.table))) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: %s %u is not a column of table %u", what, index, c.table);
    return HY_OK;
  };
  for (uint32_t g = 0; g < n_groupby; ++g) HY_TRY(check_column(groupby[g], "GROUP BY column", g));
  for (uint32_t a = 0; a < n_aggregates; ++a) {
    HY_TRY(check_column(aggregates[a].left, "the input of aggregate", a));
    if (aggregates[a].op != HY_STAR_NO_OP) HY_TRY(check_column(aggregates[a].right, "the second input of aggregate", a));
  }

  std::vector<std::unique_ptr<ColumnHandle>> done_with;   // (destroyed when this function returns, after everything below)
  struct DoneWith {
    explicit DoneWith(std::vector<std::unique_ptr<ColumnHandle>>* list) { t_plan_done_with = list; }
    ~DoneWith() { t_plan_done_with = nullptr; }
  } done_with_scope{&done_with};
  // carried[t]: base RowIDs of table t (0 = the fact table, d + 1 = dimension d) per row of the join result so far
  std::vector<std::unique_ptr<DeviceBuffer>> carried(n_dimensions + 1);
  std::vector<const hy_row_id*> carried_rows(n_dimensions + 1, nullptr);
  std::vector<std::unique_ptr<JoinOutput>> joins;   // (the PosLists a carried pointer may still point into)
  uint64_t n_rows = 0;
  // The fused probe (csrc/join_star.hpp): every dimension's filtered keys become a direct table, ONE pass over the fact table's foreign keys
  // finds the rows that have a partner in every dimension, and only those rows' RowIDs are written -- no pair lists, no RowIDs gathered
  // through them from join to join.  The rows come in the fact table's order (the chain: in the order its last JoinHash leaves them in);
  // the groups and their cells are the same, the ORDER of the groups follows the rows'.
  bool fused = false;
  t_last_star_was_fused = 0;
  {
    std::vector<std::unique_ptr<DeviceBuffer>> dimension_rows(n_dimensions), counts(n_dimensions);
    std::vector<StarProbeDimension> probes(n_dimensions);
    bool shape_ok = true;
    for (uint32_t d = 0; d < n_dimensions && shape_ok; ++d) {
      const hy_star_dimension& dimension = dimensions[d];
      if (dimension.key->is_reference || dimension.key->data_type != HY_TYPE_INT || dimension.fact_key->data_type != HY_TYPE_INT) { shape_ok = false; break; }
      dimension_rows[d] = std::make_unique<DeviceBuffer>();
      uint64_t n_dimension_rows = dimension.key->rows;
      const uint64_t* count_in_memory = nullptr;
      const hy_row_id* rows = nullptr;
      const ScanJob* filter_jobs = nullptr;
      // The kernels that build a dimension's tables walk the table itself and test its filter row by row (join_star.hpp: no TableScan, no
      // PosList for a table of a few thousand to a million rows) where its columns are plain data columns; otherwise: the rows a scan leaves.
      const bool in_place = !dimension.key->has_compressed && !dimension.key->is_mvcc &&
                            (!dimension.filter_column || (!dimension.filter_column->is_reference && !dimension.filter_column->is_mvcc && !dimension.filter_column->has_compressed));
      bool tested_in_place = in_place && !dimension.filter_column;
      if (in_place && dimension.filter_column) {
        const uint32_t n_chunks = dimension.filter_column->n_chunks;
        const size_t jobs_bytes = (sizeof(ScanJob) * (size_t{n_chunks} + 1) + 255) & ~size_t{255};
        counts[d] = std::make_unique<DeviceBuffer>();   // (the jobs and what prepare_jobs stages of the predicate)
        HY_TRY(counts[d]->alloc(jobs_bytes + scan_jobs_staging_bytes(dimension.filter_column, &dimension.predicate) + 512));
        ScanJob* jobs = counts[d]->as<ScanJob>();
        if (prepare_scan_jobs(dimension.filter_column, &dimension.predicate, jobs, counts[d]->as<char>() + jobs_bytes) == HY_OK) {
          filter_jobs = jobs;
          tested_in_place = true;
        }   // (a predicate the scan refuses: hy_table_scan below says why)
      }
      if (!tested_in_place) {
        if (dimension.filter_column) {
          counts[d] = std::make_unique<DeviceBuffer>();
          HY_TRY(filtered_rows(dimension.filter_column, &dimension.predicate, *dimension_rows[d], &n_dimension_rows, counts[d].get()));
          count_in_memory = counts[d]->as<uint64_t>() + dimension.filter_column->n_chunks;
        } else HY_TRY(star_all_rows_of(dimension.key, *dimension_rows[d]));
        rows = dimension_rows[d]->as<hy_row_id>();
      }
      bool wanted = false;
      for (uint32_t g = 0; g < n_groupby; ++g) wanted = wanted || groupby[g].table == d + 1;
      for (uint32_t a = 0; a < n_aggregates; ++a) wanted = wanted || (aggregates[a].left.column && aggregates[a].left.table == d + 1) || (aggregates[a].op != HY_STAR_NO_OP && aggregates[a].right.table == d + 1);
      probes[d] = StarProbeDimension{dimension.key, rows, filter_jobs ? dimension.filter_column : nullptr, filter_jobs, n_dimension_rows, count_in_memory, dimension.fact_key, wanted};
    }
    if (shape_ok) {
      auto fact_rows = std::make_unique<DeviceBuffer>();
      std::vector<std::unique_ptr<DeviceBuffer>> rows_of_dimension;
      // the aggregate inside the join, where its shape allows (star_finish, join_star.hpp; star_probe_rows checks the columns)
      StarFinishRequest finish;
      std::memset(&finish, 0, sizeof(finish));
      StarFinishGroups finished;
      const bool ask = result->mem == HY_MEM_HOST && n_groupby >= 1 && n_groupby <= 4 && n_aggregates <= 8;
      if (ask) {
        finish.n_groupby = n_groupby;
        finish.n_aggregates = n_aggregates;
        for (uint32_t g = 0; g < n_groupby; ++g) finish.groupby[g] = StarFinishColumnSpec{groupby[g].table, groupby[g].column};
        for (uint32_t g = 0; g < n_aggregates; ++g) {
          finish.aggregates[g].function = aggregates[g].function;
          finish.aggregates[g].op = aggregates[g].op;
          finish.aggregates[g].left = StarFinishColumnSpec{aggregates[g].left.table, aggregates[g].left.column};
          finish.aggregates[g].right = StarFinishColumnSpec{aggregates[g].right.table, aggregates[g].op != HY_STAR_NO_OP ? aggregates[g].right.column : nullptr};
        }
      }
      HY_TRY(star_probe_rows(probes.data(), n_dimensions, *fact_rows, rows_of_dimension, &n_rows, &fused, ask ? &finish : nullptr, ask ? &finished : nullptr));
      t_last_star_was_fused = fused ? (finished.done ? 2 : 1) : 0;
      if (fused && finished.done) {
        if (joined_rows) *joined_rows = n_rows;
        return write_star_groups(finished, groupby, n_groupby, aggregates, n_aggregates, n_rows, result);
      }
      if (fused) {
        carried[0] = std::move(fact_rows);
        carried_rows[0] = carried[0]->as<hy_row_id>();
        for (uint32_t d = 0; d < n_dimensions; ++d) {
          if (!rows_of_dimension[d]) continue;
          carried[d + 1] = std::move(rows_of_dimension[d]);
          carried_rows[d + 1] = carried[d + 1]->as<hy_row_id>();
        }
      } else n_rows = 0;
    }
  }
  for (uint32_t d = 0; d < n_dimensions && !fused; ++d) {
    const hy_star_dimension& dimension = dimensions[d];
    // build side: the dimension's keys, of the rows that pass its filter
    DeviceBuffer dimension_rows, build_keys;
    uint64_t n_dimension_rows = 0;
    ColumnHandle build;
    const hy_column* build_column = dimension.key;
    if (dimension.filter_column) {
      HY_TRY(filtered_rows(dimension.filter_column, &dimension.predicate, dimension_rows, &n_dimension_rows));
      HY_TRY(materialise(dimension.key, dimension_rows.as<hy_row_id>(), n_dimension_rows, build_keys, build));
      build_column = build.column;
    }
    // probe side: the fact table's foreign key, of the rows still in the join
    DeviceBuffer probe_keys;
    ColumnHandle probe;
    const hy_column* probe_column = dimension.fact_key;
    if (d > 0) {
      HY_TRY(materialise(dimension.fact_key, carried_rows[0], n_rows, probe_keys, probe));
      probe_column = probe.column;
    }
    joins.push_back(std::make_unique<JoinOutput>());
    JoinOutput& join = *joins.back();
    HY_TRY(join_inner(build_column, probe_column, join));
    const uint64_t n_pairs = join.n_pairs;
    // what every surviving row is made of
    std::vector<std::unique_ptr<DeviceBuffer>> next(n_dimensions + 1);
    std::vector<const hy_row_id*> next_rows(n_dimensions + 1, nullptr);
    if (d == 0) next_rows[0] = join.right;   // (positions in the fact table ARE its RowIDs: the probe column is the data column)
    for (uint32_t t = 0; t <= d; ++t) {
      if (d == 0 || (t > 0 && !carried_rows[t])) continue;
      next[t] = std::make_unique<DeviceBuffer>();
      HY_TRY(next[t]->alloc(sizeof(hy_row_id) * std::max<uint64_t>(1, n_pairs)));
      HY_TRY(hy_gather_row_ids(carried_rows[t], n_rows, DENSE_CHUNK, join.right, n_pairs, next[t]->as<hy_row_id>()));
      next_rows[t] = next[t]->as<hy_row_id>();
    }
    if (dimension.filter_column) {
      next[d + 1] = std::make_unique<DeviceBuffer>();
      HY_TRY(next[d + 1]->alloc(sizeof(hy_row_id) * std::max<uint64_t>(1, n_pairs)));
      HY_TRY(hy_gather_row_ids(dimension_rows.as<hy_row_id>(), n_dimension_rows, DENSE_CHUNK, join.left, n_pairs, next[d + 1]->as<hy_row_id>()));
      next_rows[d + 1] = next[d + 1]->as<hy_row_id>();
    } else next_rows[d + 1] = join.left;   // (an unfiltered dimension is joined as the data column itself)
    // The kernels that read this round's inputs are queued on this thread's stream; the buffers released here go back to its pool and are
    // handed out again in stream order.
    carried.swap(next);
    carried_rows.swap(next_rows);
    n_rows = n_pairs;
    // join outputs nobody points into any more
    for (auto& earlier : joins) {
      if (!earlier) continue;
      bool used = false;
      for (const hy_row_id* rows : carried_rows) used = used || rows == earlier->left || rows == earlier->right;
      if (!used) earlier.reset();
    }
  }
  if (joined_rows) *joined_rows = n_rows;

  // the columns the aggregate reads, materialised at the surviving rows (each (table, column) once)
  struct Output { hy_star_column source; DeviceBuffer values; ColumnHandle column; };
  std::vector<std::unique_ptr<Output>> outputs;
  auto output_of = [&](const hy_star_column& source, const hy_column** column) -> hy_status {
    for (auto& o : outputs) if (o->source.table == source.table && o->source.column == source.column) { *column = o->column.column; return HY_OK; }
    outputs.push_back(std::make_unique<Output>());
    Output& o = *outputs.back();
    o.source = source;
    if (!carried_rows[source.table]) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: no rows carried for table %u (internal error)", source.table);
    HY_TRY(materialise(source.column, carried_rows[source.table], n_rows, o.values, o.column));
    *column = o.column.column;
    return HY_OK;
  };
  std::vector<const hy_column*> groupby_columns(n_groupby ? n_groupby : 1, nullptr);
  for (uint32_t g = 0; g < n_groupby; ++g) HY_TRY(output_of(groupby[g], &groupby_columns[g]));
  std::vector<hy_aggregate_spec> specs(n_aggregates ? n_aggregates : 1);
  std::vector<std::unique_ptr<ColumnHandle>> expressions;
  for (uint32_t a = 0; a < n_aggregates; ++a) {
    specs[a].function = aggregates[a].function;
    specs[a].column = nullptr;
    if (!aggregates[a].left.column) continue;   // COUNT(*)
    const hy_column* left = nullptr;
    HY_TRY(output_of(aggregates[a].left, &left));
    if (aggregates[a].op == HY_STAR_NO_OP) { specs[a].column = left; continue; }
    const hy_column* right = nullptr;
    HY_TRY(output_of(aggregates[a].right, &right));
    hy_operand l, r;
    std::memset(&l, 0, sizeof(l));
    std::memset(&r, 0, sizeof(r));
    l.column = left;
    r.column = right;
    expressions.push_back(std::make_unique<ColumnHandle>());
    HY_TRY(hy_projection_arithmetic(aggregates[a].op, &l, &r, &expressions.back()->column));
    specs[a].column = expressions.back()->column;
  }
  if (n_groupby == 0 && n_aggregates == 0) return fail(HY_ERR_INVALID, "hy_star_join_aggregate: nothing to aggregate");
  // The aggregate's input columns are new in every call: which path this GROUP BY should start on (hy_device.hpp) is remembered with the
  // caller's first GROUP BY column -- a data column that stays -- under the signature of the plan's columns.
  uint64_t signature = 0x51A9E5B1ull * (n_groupby + 1) + n_dimensions;
  for (uint32_t g = 0; g < n_groupby; ++g) signature = (signature ^ reinterpret_cast<uintptr_t>(groupby[g].column)) * 0xD6E8FEB86659FD93ull;
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    const uint64_t literal = dimensions[d].filter_column ? static_cast<uint64_t>(dimensions[d].predicate.value.i64) ^ (uint64_t{dimensions[d].predicate.condition} << 56) : 0;   // (no filter: the predicate is not read)
    signature = (signature ^ reinterpret_cast<uintptr_t>(dimensions[d].key) ^ literal) * 0xD6E8FEB86659FD93ull;
  }
  signature |= 0x100;
  const hy_column* hint_owner = n_groupby ? groupby[0].column : nullptr;
  if (hint_owner) {
    const uint64_t hint = hint_owner->star_aggregate_hint.load(std::memory_order_relaxed);
    if (hint >> 8 == signature >> 8) t_aggregate_next_path = static_cast<uint32_t>(hint & 0xFF);
  }
  const hy_status status = hy_aggregate_hash(groupby_columns.data(), n_groupby, specs.data(), n_aggregates, result);
  t_aggregate_next_path = 0;   // (a call that returned before it consumed the hint must not leave it to the thread's next aggregate)
  if (status == HY_OK && hint_owner && n_groupby <= 4) hint_owner->star_aggregate_hint.store((signature >> 8) << 8 | (t_aggregate_recommended & 0xFF), std::memory_order_relaxed);
  return status;
}

// debug / tests only: 1 = the calling thread's last hy_star_join_aggregate probed every dimension in one pass (csrc/join_star.hpp), 2 = and grouped
// the survivors inside that pass (star_finish)
int hy_debug_star_fused(void) { return t_last_star_was_fused; }

}  // extern "C"
