// multi_gpu.hpp -- ONE process, several GPUs: the C++ coordinator of the sharded operators.
//
// Hyrise is a single process whose operators run on scheduler workers (scheduler/operator_task.cpp:163-200,
// abstract_scheduler.cpp:53-63: one task per operator, one job per chunk).  The multi-GPU shape that fits it is one worker thread per
// GPU inside that process: every worker is bound to its device (hy_bind_device), holds the chunk range [first, last) of every column it
// was given (device_column_of_chunks) and one RCCL communicator (hy_comm_init_all = ncclCommInitAll, RCCL's single-process mode).
// TableScans need no exchange (chunks are independent); this file holds the three operators that do:
//   sharded_aggregate          per-rank hy_aggregate_hash, partial aggregates merged over a fixed slot table with ncclAllReduce when the
//                              GROUP BY keys span a small integer range (TPC-H Q1: 2 x 3 values), else the ranks' (key, partial) tables
//                              travel with ncclAllGather and every rank merges them in rank order
//   sharded_join_broadcast     the build column's shards are gathered on every GPU (ncclAllGather), every rank joins its probe shard
//   sharded_join_repartition   both sides' (key, RowID) tuples go to rank key % G -- JoinHash's own radix function, the integer itself
//                              (join_hash_steps.hpp:352, fan-out :411-417, :593-599) -- with one grouped ncclSend / ncclRecv per side
// Counts are exchanged ONCE per collective of variable size (an 8-byte-per-rank all-gather, read back with the single host
// synchronisation that sizes the receive buffers); all data stays in HBM between the collective and the operator that consumes it.
// hyrise_amd/distributed.py is the same coordinator over torch.distributed for the one-process-per-GPU launch of bench.py and for the
// gloo tests on CPU; results are identical by construction (same partition function, same merge order).
#pragma once

#include <condition_variable>
#include <exception>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>

#include "hyrise_host.hpp"

namespace hyrise_amd {

// ---- the workers ---------------------------------------------------------------------------------------------------------------
class DeviceGroup {
 public:
  explicit DeviceGroup(std::vector<int32_t> devices) : _devices(std::move(devices)), _comms(_devices.size(), nullptr) {
    Assert(!_devices.empty(), "DeviceGroup: no device");
    check_status(hy_comm_init_all(_devices.data(), static_cast<uint32_t>(_devices.size()), _comms.data()));
    for (uint32_t rank = 0; rank < size(); ++rank) _workers.emplace_back([this, rank] { _work(rank); });
    run([](uint32_t, hy_comm*) {});   // every worker bound (or the binding's error surfaces here)
  }
  ~DeviceGroup() {
    {
      std::lock_guard<std::mutex> lock(_mutex);
      _stop = true;
      ++_generation;
    }
    _wake.notify_all();
    for (auto& worker : _workers) worker.join();
    for (auto* comm : _comms) hy_comm_destroy(comm);
  }
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;

  uint32_t size() const { return static_cast<uint32_t>(_devices.size()); }
  int32_t device(uint32_t rank) const { return _devices[rank]; }

  // task(rank, communicator) on every worker at once; returns when all are done.  The first exception of a worker is rethrown here --
  // after ALL workers have returned: a worker that fails before a collective would leave the others waiting in it, so tasks validate
  // their arguments (identically on every rank) before the first exchange.
  void run(const std::function<void(uint32_t, hy_comm*)>& task) {
    std::unique_lock<std::mutex> lock(_mutex);
    _task = &task;
    _pending = size();
    _failure = nullptr;
    ++_generation;
    _wake.notify_all();
    _done.wait(lock, [this] { return _pending == 0; });
    _task = nullptr;
    if (_failure) std::rethrow_exception(_failure);
  }

 private:
  void _work(uint32_t rank) {
    const hy_status bound = hy_bind_device(_devices[rank]);
    const std::string bind_error = bound == HY_OK ? "" : hy_last_error();
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(uint32_t, hy_comm*)>* task = nullptr;
      {
        std::unique_lock<std::mutex> lock(_mutex);
        _wake.wait(lock, [&] { return _generation != seen; });
        seen = _generation;
        if (_stop) return;
        task = _task;
      }
      std::exception_ptr failure;
      try {
        if (bound != HY_OK) Fail("DeviceGroup worker: " + bind_error);
        (*task)(rank, _comms[rank]);
        check_status(hy_synchronize());
      } catch (...) {
        failure = std::current_exception();
      }
      std::lock_guard<std::mutex> lock(_mutex);
      if (failure && !_failure) _failure = failure;
      if (--_pending == 0) _done.notify_all();
    }
  }

  std::vector<int32_t> _devices;
  std::vector<hy_comm*> _comms;
  std::vector<std::thread> _workers;
  std::mutex _mutex;
  std::condition_variable _wake, _done;
  const std::function<void(uint32_t, hy_comm*)>* _task = nullptr;
  uint64_t _generation = 0;
  uint32_t _pending = 0;
  bool _stop = false;
  std::exception_ptr _failure;
};

// The chunks [begin, end) of a table of n_chunks chunks that rank `rank` of `world` holds (hyrise_amd/distributed.py chunk_range).
inline std::pair<ChunkID, ChunkID> chunk_range(ChunkID n_chunks, uint32_t world, uint32_t rank) {
  const ChunkID base = n_chunks / world, extra = n_chunks % world;
  const ChunkID begin = rank * base + std::min<ChunkID>(rank, extra);
  return {begin, begin + base + (rank < extra ? 1 : 0)};
}

// One column of a table, split by chunk range over the workers of a group (built ON the workers: every shard lives on its device).
struct ShardedColumn {
  std::vector<std::shared_ptr<DeviceColumn>> shard;   // [world]
  std::vector<ChunkID> first_chunk;                   // [world] the shard's first chunk in the whole table
  DataType data_type = DataType::Int;
  ChunkOffset chunk_rows = Chunk::DEFAULT_SIZE;       // target chunk size of the table (all chunks but the last)
};

inline ShardedColumn shard_column(DeviceGroup& group, const std::shared_ptr<const Table>& table, ColumnID column_id, StringKeys string_keys = StringKeys::None) {
  ShardedColumn out;
  out.shard.resize(group.size());
  out.first_chunk.resize(group.size());
  out.data_type = table->column_data_type(column_id);
  out.chunk_rows = table->target_chunk_size();
  group.run([&](uint32_t rank, hy_comm*) {
    const auto [begin, end] = chunk_range(table->chunk_count(), group.size(), rank);
    out.first_chunk[rank] = begin;
    out.shard[rank] = device_column_of_chunks(table, column_id, string_keys, begin, end);
  });
  return out;
}

// ---- device memory of a worker ----------------------------------------------------------------------------------------------
class DeviceBytes {
 public:
  DeviceBytes() = default;
  explicit DeviceBytes(size_t bytes) { check_status(hy_device_malloc(&_ptr, std::max<size_t>(bytes, 16))); }
  ~DeviceBytes() { if (_ptr) hy_device_free(_ptr); }
  DeviceBytes(DeviceBytes&& other) noexcept : _ptr(other._ptr) { other._ptr = nullptr; }
  DeviceBytes& operator=(DeviceBytes&& other) noexcept { std::swap(_ptr, other._ptr); return *this; }
  DeviceBytes(const DeviceBytes&) = delete;
  DeviceBytes& operator=(const DeviceBytes&) = delete;
  void* get() const { return _ptr; }
  template <typename T> T* as() const { return static_cast<T*>(_ptr); }

 private:
  void* _ptr = nullptr;
};

namespace detail {

inline uint64_t column_rows(const hy_column* column) {
  uint64_t rows = 0;
  check_status(hy_column_row_count(column, &rows));
  return rows;
}

// Every rank's `mine`: one 8-byte-per-rank all-gather, then the only host synchronisation of a variable-size exchange.
inline std::vector<uint64_t> all_gather_counts(hy_comm* comm, uint32_t world, const uint64_t* mine, uint32_t per_rank) {
  DeviceBytes send(per_rank * sizeof(uint64_t)), recv(size_t{world} * per_rank * sizeof(uint64_t));
  check_status(hy_memcpy_h2d(send.get(), mine, per_rank * sizeof(uint64_t)));
  check_status(hy_comm_all_gather(comm, send.get(), recv.get(), per_rank * sizeof(uint64_t)));
  std::vector<uint64_t> all(size_t{world} * per_rank);
  check_status(hy_memcpy_d2h(all.data(), recv.get(), all.size() * sizeof(uint64_t)));   // (stream-ordered behind the collective)
  return all;
}

// A column over `rows` values of `width` bytes at `values` (device memory, nothing copied): chunks of chunk_rows rows.
inline hy_column* device_value_column(const void* values, uint64_t rows, uint32_t width, uint32_t data_type, uint32_t chunk_rows,
                                      const std::vector<const void*>* chunk_values = nullptr, const std::vector<uint64_t>* chunk_sizes = nullptr,
                                      const std::vector<const void*>* chunk_nulls = nullptr) {
  std::vector<hy_segment> segments;
  const auto add = [&](const void* data, uint64_t size, const void* nulls) {
    hy_segment s{};
    s.encoding = HY_ENC_UNENCODED; s.data_type = data_type; s.size = static_cast<uint32_t>(size); s.width = width;
    s.data = data; s.nulls = static_cast<const uint64_t*>(nulls); s.ref_chunk_id = 0xFFFFFFFFu;
    segments.push_back(s);
  };
  if (chunk_values) {
    for (size_t c = 0; c < chunk_values->size(); ++c) add((*chunk_values)[c], (*chunk_sizes)[c], chunk_nulls ? (*chunk_nulls)[c] : nullptr);
  } else {
    for (uint64_t begin = 0; begin < rows; begin += chunk_rows) add(static_cast<const char*>(values) + begin * width, std::min<uint64_t>(chunk_rows, rows - begin), nullptr);
  }
  hy_column* column = nullptr;
  check_status(hy_column_create(segments.data(), static_cast<uint32_t>(segments.size()), HY_MEM_DEVICE, &column));
  return column;
}

struct ColumnHandle {   // hy_column_destroy on scope exit
  hy_column* handle = nullptr;
  ~ColumnHandle() { if (handle) hy_column_destroy(handle); }
};

// hy_join_hash with the PosLists in device memory; -> pairs.  left_positions / right_positions are (re)allocated here.
inline uint64_t device_join(const hy_column* left, const hy_column* right, JoinMode mode, DeviceBytes& left_positions, DeviceBytes& right_positions) {
  uint32_t left_chunks = 0, right_chunks = 0;
  check_status(hy_column_chunk_count(left, &left_chunks));
  check_status(hy_column_chunk_count(right, &right_chunks));
  uint64_t capacity = std::max<uint64_t>(1, std::max(column_rows(left), column_rows(right)));
  uint32_t slice_capacity = static_cast<uint32_t>(capacity / 131070 + std::max(left_chunks, right_chunks) + 600);
  for (int attempt = 0;; ++attempt) {
    left_positions = DeviceBytes(capacity * sizeof(hy_row_id));
    right_positions = DeviceBytes(capacity * sizeof(hy_row_id));
    DeviceBytes slice_offsets((size_t{slice_capacity} + 2) * sizeof(uint64_t));
    hy_join_result result{};
    result.mem = HY_MEM_DEVICE;
    result.radix_bits = 0xFFFFFFFFu;
    result.left_pos = left_positions.as<hy_row_id>();
    result.right_pos = right_positions.as<hy_row_id>();
    result.capacity = capacity;
    result.slice_offsets = slice_offsets.as<uint64_t>();
    result.slice_capacity = slice_capacity;
    const auto status = hy_join_hash(left, right, static_cast<uint32_t>(mode), &result);
    if (status == HY_ERR_CAPACITY && attempt == 0 && (result.n_pairs > capacity || result.n_slices > slice_capacity)) {
      capacity = std::max<uint64_t>(capacity, result.n_pairs);
      slice_capacity = std::max(slice_capacity, result.n_slices);
      continue;
    }
    check_status(status);
    check_status(hy_synchronize());   // (slice_offsets is freed on return)
    return result.n_pairs;
  }
}

inline std::vector<RowID> read_row_ids(const void* device, uint64_t n, uint32_t add_to_chunk_id = 0) {
  std::vector<RowID> rows(n);
  if (n) check_status(hy_memcpy_d2h(rows.data(), device, n * sizeof(RowID)));
  if (add_to_chunk_id) for (auto& row : rows) if (!row.is_null()) row.chunk_id += add_to_chunk_id;
  return rows;
}

inline bool is_semi_or_anti(JoinMode mode) { return mode == JoinMode::Semi || mode == JoinMode::AntiNullAsTrue || mode == JoinMode::AntiNullAsFalse; }

}  // namespace detail

// RowIDs of the WHOLE tables, per rank (the union over the ranks is the join's result; pairs of different ranks in rank order).
struct ShardedJoinOutput {
  std::vector<std::vector<RowID>> left, right;   // [world]; right stays empty for Semi
};

// ---- hash repartition ---------------------------------------------------------------------------------------------------------
// Inner and Semi (the modes whose rows with NULL keys vanish: join_hash.cpp:284-286 keeps them for the outer / anti modes, which
// hyrise_amd/distributed.py sharded_join_repartition adds from the rank that holds them).
inline ShardedJoinOutput sharded_join_repartition(DeviceGroup& group, const ShardedColumn& left, const ShardedColumn& right, JoinMode mode = JoinMode::Inner) {
  Assert(mode == JoinMode::Inner || mode == JoinMode::Semi, "sharded_join_repartition (C++): Inner and Semi");
  Assert(left.data_type == right.data_type && (left.data_type == DataType::Int || left.data_type == DataType::Long), "hash repartition: two int or two long join columns");
  const uint32_t world = group.size();
  const uint32_t key_width = left.data_type == DataType::Int ? 4 : 8;
  constexpr uint32_t RECEIVED_CHUNK = 65535;   // received tuple arrays are presented as columns of Chunk::DEFAULT_SIZE rows
  ShardedJoinOutput out;
  out.left.resize(world);
  out.right.resize(world);
  group.run([&](uint32_t rank, hy_comm* comm) {
    struct Side { DeviceBytes keys, rows; uint64_t n = 0; detail::ColumnHandle column; };
    Side sides[2];
    for (int s = 0; s < 2; ++s) {
      const ShardedColumn& input = s == 0 ? left : right;
      const hy_column* column = input.shard[rank]->handle;
      const uint64_t rows = detail::column_rows(column);
      DeviceBytes keys(rows * key_width), row_ids(rows * sizeof(hy_row_id));
      std::vector<uint64_t> send_tuples(world, 0);
      check_status(hy_repartition_pack(column, world, input.first_chunk[rank], keys.get(), row_ids.as<hy_row_id>(), rows, send_tuples.data()));
      const auto counts = detail::all_gather_counts(comm, world, send_tuples.data(), world);   // counts[p * world + q]: tuples p sends to q
      std::vector<uint64_t> send_keys(world), send_rows(world), recv_keys(world), recv_rows(world);
      for (uint32_t peer = 0; peer < world; ++peer) {
        const uint64_t incoming = counts[size_t{peer} * world + rank];
        send_keys[peer] = send_tuples[peer] * key_width; send_rows[peer] = send_tuples[peer] * sizeof(hy_row_id);
        recv_keys[peer] = incoming * key_width; recv_rows[peer] = incoming * sizeof(hy_row_id);
        sides[s].n += incoming;
      }
      sides[s].keys = DeviceBytes(sides[s].n * key_width);
      sides[s].rows = DeviceBytes(sides[s].n * sizeof(hy_row_id));
      check_status(hy_comm_all_to_all_v(comm, keys.get(), send_keys.data(), sides[s].keys.get(), recv_keys.data()));
      check_status(hy_comm_all_to_all_v(comm, row_ids.get(), send_rows.data(), sides[s].rows.get(), recv_rows.data()));
      check_status(hy_synchronize());   // the send buffers go out of scope
      sides[s].column.handle = detail::device_value_column(sides[s].keys.get(), sides[s].n, key_width, static_cast<uint32_t>(input.data_type), RECEIVED_CHUNK);
    }
    DeviceBytes left_positions, right_positions;
    const uint64_t pairs = detail::device_join(sides[0].column.handle, sides[1].column.handle, mode, left_positions, right_positions);
    DeviceBytes translated(pairs * sizeof(hy_row_id));
    if (pairs) check_status(hy_gather_row_ids(sides[0].rows.as<hy_row_id>(), sides[0].n, RECEIVED_CHUNK, left_positions.as<hy_row_id>(), pairs, translated.as<hy_row_id>()));
    out.left[rank] = detail::read_row_ids(translated.get(), pairs);
    if (mode == JoinMode::Inner) {
      if (pairs) check_status(hy_gather_row_ids(sides[1].rows.as<hy_row_id>(), sides[1].n, RECEIVED_CHUNK, right_positions.as<hy_row_id>(), pairs, translated.as<hy_row_id>()));
      out.right[rank] = detail::read_row_ids(translated.get(), pairs);
    }
  });
  return out;
}

// ---- broadcast build -----------------------------------------------------------------------------------------------------------
// `build` is gathered on every GPU and must be the side hy_join_hash builds on (join_hash.cpp:139-155: the right input of Left / Semi /
// Anti*, the left input of Right, either side of Inner) -- as the probe or outer side it would be emitted once per rank.  The build
// column must not hold NULLs here (keys of a dimension table; hyrise_amd/distributed.py carries the null bytes along as well).
inline ShardedJoinOutput sharded_join_broadcast(DeviceGroup& group, const ShardedColumn& build, const ShardedColumn& probe, JoinMode mode, bool build_is_left) {
  const bool allowed = mode == JoinMode::Inner || (mode == JoinMode::Right ? build_is_left : (!build_is_left && (mode == JoinMode::Left || detail::is_semi_or_anti(mode))));
  Assert(allowed, "sharded_join_broadcast: the gathered column must be the side the join builds on");
  const uint32_t world = group.size();
  const uint32_t width = (build.data_type == DataType::Int || build.data_type == DataType::Float) ? 4 : 8;
  Assert(build.data_type != DataType::String && probe.data_type != DataType::String, "sharded_join_broadcast: numeric join columns (strings join as ids, INTEGRATION.md section 3)");
  ShardedJoinOutput out;
  out.left.resize(world);
  out.right.resize(world);
  group.run([&](uint32_t rank, hy_comm* comm) {
    const hy_column* mine = build.shard[rank]->handle;
    const uint64_t rows = detail::column_rows(mine);
    uint32_t my_chunks = 0;
    check_status(hy_column_chunk_count(mine, &my_chunks));
    const uint64_t my_shape[2] = {rows, my_chunks};
    const auto shapes = detail::all_gather_counts(comm, world, my_shape, 2);
    uint64_t most = 0;
    for (uint32_t peer = 0; peer < world; ++peer) most = std::max(most, shapes[2 * peer]);
    // ncclAllGather moves equal pieces: every rank contributes `most` rows (the tail of a shorter shard is never read -- the gathered
    // column's segments point at each rank's piece, whole chunks of the build table in rank order)
    DeviceBytes piece(most * width), null_bytes(std::max<uint64_t>(rows, 1)), gathered(size_t{world} * most * width);
    check_status(hy_column_export(mine, piece.get(), null_bytes.as<uint8_t>()));
    std::vector<uint8_t> nulls(rows);
    if (rows) check_status(hy_memcpy_d2h(nulls.data(), null_bytes.get(), rows));
    Assert(std::find(nulls.begin(), nulls.end(), uint8_t{1}) == nulls.end(), "sharded_join_broadcast (C++): NULL keys on the build side");
    check_status(hy_comm_all_gather(comm, piece.get(), gathered.get(), most * width));
    std::vector<const void*> chunk_values;
    std::vector<uint64_t> chunk_sizes;
    for (uint32_t peer = 0; peer < world; ++peer) {
      const uint64_t peer_rows = shapes[2 * peer];
      for (uint64_t begin = 0; begin < peer_rows; begin += build.chunk_rows) {
        chunk_values.push_back(gathered.as<char>() + (size_t{peer} * most + begin) * width);
        chunk_sizes.push_back(std::min<uint64_t>(build.chunk_rows, peer_rows - begin));
      }
      Assert(peer + 1 == world || peer_rows % build.chunk_rows == 0, "sharded_join_broadcast: shards are whole chunks of the build table");
    }
    detail::ColumnHandle whole;
    whole.handle = detail::device_value_column(nullptr, 0, width, static_cast<uint32_t>(build.data_type), build.chunk_rows, &chunk_values, &chunk_sizes);
    const hy_column* probe_column = probe.shard[rank]->handle;
    DeviceBytes left_positions, right_positions;
    const uint64_t pairs = build_is_left ? detail::device_join(whole.handle, probe_column, mode, left_positions, right_positions)
                                         : detail::device_join(probe_column, whole.handle, mode, left_positions, right_positions);
    const uint32_t probe_first = probe.first_chunk[rank];
    out.left[rank] = detail::read_row_ids(left_positions.get(), pairs, build_is_left ? 0 : probe_first);
    if (!detail::is_semi_or_anti(mode)) out.right[rank] = detail::read_row_ids(right_positions.get(), pairs, build_is_left ? probe_first : 0);
  });
  return out;
}

// ---- sharded AggregateHash ---------------------------------------------------------------------------------------------------
// MIN / MAX / SUM / AVG / COUNT over numeric columns, GROUP BY integer columns (int / long; string keys of four bytes or fewer arrive
// as their AggregateKey names, StringKeys::AggregateKeyNames).  Every rank computes the same merged table; rank 0's copy is returned.
struct ShardedAggregate {
  WindowFunction function = WindowFunction::Count;
  const ShardedColumn* column = nullptr;   // nullptr: COUNT(*)
};
struct MergedGroup {
  std::vector<std::optional<int64_t>> key;        // one entry per GROUP BY column (nullopt: the NULL group)
  std::vector<std::optional<double>> value;       // one entry per aggregate (nullopt: NULL); integers are exact below 2^53 ...
  std::vector<int64_t> integer;                   // ... and carried exactly here for integer results (0 for floating-point ones)
  uint64_t first_row = 0;                         // (chunk id in the whole table) << 32 | chunk offset of the group's first row
};
struct ShardedAggregateOutput {
  std::vector<MergedGroup> groups;   // the single-process operator's group order: first occurrence in the table
  bool used_all_reduce = false;      // the fixed-slot path ran (keys of a small integer range)
};

namespace detail {

struct LocalGroups {   // one rank's hy_aggregate_hash: per group the key values, the first row and per aggregate (value, count)
  uint32_t n = 0;
  std::vector<hy_row_id> first_row;
  std::vector<std::vector<int64_t>> key;          // [groupby][group]
  std::vector<std::vector<uint8_t>> key_null;
  std::vector<std::vector<double>> value;         // [aggregate][group] the partial: SUM / MIN / MAX as a double ...
  std::vector<std::vector<int64_t>> integer;      // ... and exactly, for integer results
  std::vector<std::vector<int64_t>> count;        // [aggregate][group] non-NULL inputs (COUNT(*): rows)
};

inline bool result_is_integer(uint32_t data_type) { return data_type == HY_TYPE_INT || data_type == HY_TYPE_LONG; }

inline LocalGroups local_groups(const std::vector<const hy_column*>& groupby, const std::vector<std::pair<uint32_t, const hy_column*>>& partials) {
  // the plan: ANY of every GROUP BY column (the key values), then per aggregate its partial and the COUNT of its column
  std::vector<hy_aggregate_spec> plan;
  for (const auto* column : groupby) plan.push_back(hy_aggregate_spec{HY_AGG_ANY, column});
  for (const auto& [function, column] : partials) {
    plan.push_back(hy_aggregate_spec{function, column});
    plan.push_back(hy_aggregate_spec{HY_AGG_COUNT, column});
  }
  LocalGroups out;
  out.key.resize(groupby.size()); out.key_null.resize(groupby.size());
  out.value.resize(partials.size()); out.integer.resize(partials.size()); out.count.resize(partials.size());
  const hy_column* shape = !groupby.empty() ? groupby[0] : nullptr;
  for (const auto& partial : partials) if (!shape) shape = partial.second;
  if (!shape || column_rows(shape) == 0) return out;
  constexpr size_t PER_CALL = 8;   // hy_aggregate_hash's limit; every call groups the same rows in the same order
  uint32_t capacity = 1u << 16;
  for (size_t begin = 0; begin < plan.size() || begin == 0; begin += PER_CALL) {
    const size_t n_specs = std::min(PER_CALL, plan.size() - begin);
    for (;;) {
      std::vector<hy_row_id> rows(capacity);
      std::vector<std::vector<uint64_t>> values(n_specs, std::vector<uint64_t>(capacity));
      std::vector<std::vector<uint8_t>> nulls(n_specs, std::vector<uint8_t>(capacity));
      std::vector<hy_aggregate_column> columns(std::max<size_t>(1, n_specs));
      for (size_t a = 0; a < n_specs; ++a) { columns[a].values = values[a].data(); columns[a].is_null = nulls[a].data(); }
      hy_aggregate_result result{};
      result.mem = HY_MEM_HOST; result.group_capacity = capacity; result.group_row_ids = rows.data(); result.columns = columns.data();
      const auto status = hy_aggregate_hash(groupby.data(), static_cast<uint32_t>(groupby.size()), plan.data() + begin, static_cast<uint32_t>(n_specs), &result);
      if (status == HY_ERR_CAPACITY && result.n_groups > capacity) { capacity = result.n_groups; continue; }
      check_status(status);
      if (begin == 0) { out.n = result.n_groups; out.first_row.assign(rows.begin(), rows.begin() + out.n); }
      Assert(result.n_groups == out.n, "hy_aggregate_hash: the calls of one plan disagree on the groups");
      for (size_t a = 0; a < n_specs; ++a) {
        const size_t cell = begin + a;
        std::vector<double> as_double(out.n);
        std::vector<int64_t> as_integer(out.n, 0);
        for (uint32_t g = 0; g < out.n; ++g) {
          switch (columns[a].data_type) {
            case HY_TYPE_INT: as_integer[g] = reinterpret_cast<const int32_t*>(values[a].data())[g]; as_double[g] = static_cast<double>(as_integer[g]); break;
            case HY_TYPE_LONG: as_integer[g] = reinterpret_cast<const int64_t*>(values[a].data())[g]; as_double[g] = static_cast<double>(as_integer[g]); break;
            case HY_TYPE_FLOAT: as_double[g] = reinterpret_cast<const float*>(values[a].data())[g]; break;
            default: as_double[g] = reinterpret_cast<const double*>(values[a].data())[g]; break;
          }
        }
        if (cell < groupby.size()) {
          Assert(result_is_integer(columns[a].data_type), "sharded_aggregate (C++): GROUP BY columns of integer type");
          out.key[cell] = as_integer;
          out.key_null[cell].assign(nulls[a].begin(), nulls[a].begin() + out.n);
        } else if ((cell - groupby.size()) % 2 == 0) {
          out.value[(cell - groupby.size()) / 2] = as_double;
          out.integer[(cell - groupby.size()) / 2] = as_integer;
        } else {
          out.count[(cell - groupby.size()) / 2] = as_integer;
        }
      }
      break;
    }
    if (plan.empty()) break;
  }
  return out;
}

}  // namespace detail

inline ShardedAggregateOutput sharded_aggregate(DeviceGroup& group, const std::vector<const ShardedColumn*>& groupby, const std::vector<ShardedAggregate>& aggregates,
                                                bool allow_all_reduce = true) {
  const uint32_t world = group.size();
  const size_t n_keys = groupby.size(), n_aggregates = aggregates.size();
  for (const auto& aggregate : aggregates) {
    const bool mergeable = aggregate.function == WindowFunction::Min || aggregate.function == WindowFunction::Max || aggregate.function == WindowFunction::Sum ||
                           aggregate.function == WindowFunction::Avg || aggregate.function == WindowFunction::Count;
    Assert(mergeable, "sharded_aggregate (C++): MIN / MAX / SUM / AVG / COUNT (hyrise_amd/distributed.py splits STDDEV_SAMP and COUNT DISTINCT into mergeable parts)");
    Assert(aggregate.column || aggregate.function == WindowFunction::Count, "Only COUNT may have an invalid ColumnID.");
    Assert(!aggregate.column || aggregate.column->data_type != DataType::String, "sharded_aggregate (C++): numeric aggregate columns");
  }
  std::vector<ShardedAggregateOutput> per_rank(world);
  constexpr uint64_t MAX_SLOTS = 4096;
  constexpr int64_t NO_ROW = std::numeric_limits<int64_t>::max();
  group.run([&](uint32_t rank, hy_comm* comm) {
    std::vector<const hy_column*> keys;
    for (const auto* column : groupby) keys.push_back(column->shard[rank]->handle);
    std::vector<std::pair<uint32_t, const hy_column*>> partials;
    std::vector<bool> integer_result(n_aggregates);
    for (size_t a = 0; a < n_aggregates; ++a) {
      const auto function = aggregates[a].function == WindowFunction::Avg ? WindowFunction::Sum : aggregates[a].function;
      partials.emplace_back(static_cast<uint32_t>(function), aggregates[a].column ? aggregates[a].column->shard[rank]->handle : nullptr);
      const bool integer_column = !aggregates[a].column || aggregates[a].column->data_type == DataType::Int || aggregates[a].column->data_type == DataType::Long;
      integer_result[a] = integer_column;
    }
    const auto local = detail::local_groups(keys, partials);
    const uint32_t first_chunk = n_keys ? groupby[0]->first_chunk[rank] : (n_aggregates && aggregates[0].column ? aggregates[0].column->first_chunk[rank] : 0);
    const auto global_row = [&](uint32_t g) { return static_cast<int64_t>((uint64_t{local.first_row[g].chunk_id} + first_chunk) << 32 | local.first_row[g].chunk_offset); };
    auto& mine = per_rank[rank];

    // -- do the keys of all ranks fit a small slot table?  per key column: min / max over the ranks (two all-reduces), one extra value for NULL
    std::vector<int64_t> low(n_keys, std::numeric_limits<int64_t>::max()), high(n_keys, std::numeric_limits<int64_t>::min());
    for (size_t k = 0; k < n_keys; ++k) {
      for (uint32_t g = 0; g < local.n; ++g) {
        if (local.key_null[k][g]) continue;
        low[k] = std::min(low[k], local.key[k][g]);
        high[k] = std::max(high[k], local.key[k][g]);
      }
    }
    const auto reduce = [&](std::vector<int64_t>& cells, uint32_t op) {
      if (cells.empty()) return;
      DeviceBytes buffer(cells.size() * sizeof(int64_t));
      check_status(hy_memcpy_h2d(buffer.get(), cells.data(), cells.size() * sizeof(int64_t)));
      check_status(hy_comm_all_reduce(comm, buffer.get(), buffer.get(), cells.size(), HY_TYPE_LONG, op));
      check_status(hy_memcpy_d2h(cells.data(), buffer.get(), cells.size() * sizeof(int64_t)));
    };
    const auto reduce_doubles = [&](std::vector<double>& cells, uint32_t op) {
      if (cells.empty()) return;
      DeviceBytes buffer(cells.size() * sizeof(double));
      check_status(hy_memcpy_h2d(buffer.get(), cells.data(), cells.size() * sizeof(double)));
      check_status(hy_comm_all_reduce(comm, buffer.get(), buffer.get(), cells.size(), HY_TYPE_DOUBLE, op));
      check_status(hy_memcpy_d2h(cells.data(), buffer.get(), cells.size() * sizeof(double)));
    };
    reduce(low, HY_COMM_MIN);
    reduce(high, HY_COMM_MAX);
    uint64_t slots = 1;
    std::vector<uint64_t> span(n_keys, 1);
    for (size_t k = 0; k < n_keys && slots <= MAX_SLOTS; ++k) {
      span[k] = low[k] > high[k] ? 1 : static_cast<uint64_t>(high[k] - low[k]) + 2;   // + the NULL group
      if (low[k] <= high[k] && static_cast<uint64_t>(high[k] - low[k]) > MAX_SLOTS) { slots = MAX_SLOTS + 1; break; }
      slots *= span[k];
    }
    const auto finish = [&](MergedGroup& merged, size_t a, double value, int64_t integer, int64_t count) {
      const auto function = aggregates[a].function;
      if (function == WindowFunction::Count) { merged.value[a] = static_cast<double>(count); merged.integer[a] = count; return; }
      if (count == 0) return;   // NULL: a group that saw only NULLs
      if (function == WindowFunction::Avg) { merged.value[a] = (integer_result[a] ? static_cast<double>(integer) : value) / static_cast<double>(count); return; }
      merged.value[a] = integer_result[a] ? static_cast<double>(integer) : value;
      merged.integer[a] = integer_result[a] ? integer : 0;
    };

    if (allow_all_reduce && slots <= MAX_SLOTS) {
      // -- fixed slots: slot = mixed-radix number of (key - low, or span - 1 for NULL); sums and counts add, MIN / MAX and the first rows reduce
      mine.used_all_reduce = true;
      std::vector<int64_t> first(slots, NO_ROW), counts(slots * n_aggregates, 0), integer_sums(slots * n_aggregates, 0);
      std::vector<int64_t> integer_min(slots * n_aggregates, std::numeric_limits<int64_t>::max()), integer_max(slots * n_aggregates, std::numeric_limits<int64_t>::min());
      std::vector<double> sums(slots * n_aggregates, 0.0), minima(slots * n_aggregates, std::numeric_limits<double>::infinity()), maxima(slots * n_aggregates, -std::numeric_limits<double>::infinity());
      for (uint32_t g = 0; g < local.n; ++g) {
        uint64_t slot = 0;
        for (size_t k = 0; k < n_keys; ++k) slot = slot * span[k] + (local.key_null[k][g] ? span[k] - 1 : static_cast<uint64_t>(local.key[k][g] - low[k]));
        first[slot] = global_row(g);
        for (size_t a = 0; a < n_aggregates; ++a) {
          const size_t cell = slot * n_aggregates + a;
          counts[cell] = local.count[a][g];
          if (!local.count[a][g] && aggregates[a].function != WindowFunction::Count) continue;
          integer_sums[cell] = integer_min[cell] = integer_max[cell] = local.integer[a][g];
          sums[cell] = minima[cell] = maxima[cell] = local.value[a][g];
        }
      }
      reduce(first, HY_COMM_MIN);
      reduce(counts, HY_COMM_SUM);
      reduce(integer_sums, HY_COMM_SUM);
      reduce(integer_min, HY_COMM_MIN);
      reduce(integer_max, HY_COMM_MAX);
      reduce_doubles(sums, HY_COMM_SUM);
      reduce_doubles(minima, HY_COMM_MIN);
      reduce_doubles(maxima, HY_COMM_MAX);
      std::vector<uint64_t> order;
      for (uint64_t slot = 0; slot < slots; ++slot) if (first[slot] != NO_ROW) order.push_back(slot);
      std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return first[a] < first[b]; });
      if (n_keys == 0 && order.empty()) order.push_back(0);   // no GROUP BY, no rows: one row of NULLs / zero counts (aggregate_hash.cpp:1422-1432)
      for (const uint64_t slot : order) {
        MergedGroup merged;
        merged.first_row = first[slot] == NO_ROW ? 0 : static_cast<uint64_t>(first[slot]);
        merged.key.resize(n_keys); merged.value.resize(n_aggregates); merged.integer.assign(n_aggregates, 0);
        uint64_t rest = slot;
        for (size_t k = n_keys; k-- > 0;) {
          const uint64_t digit = rest % span[k];
          rest /= span[k];
          if (digit != span[k] - 1) merged.key[k] = low[k] + static_cast<int64_t>(digit);
        }
        for (size_t a = 0; a < n_aggregates; ++a) {
          const size_t cell = slot * n_aggregates + a;
          const bool is_min = aggregates[a].function == WindowFunction::Min, is_max = aggregates[a].function == WindowFunction::Max;
          finish(merged, a, is_min ? minima[cell] : is_max ? maxima[cell] : sums[cell], is_min ? integer_min[cell] : is_max ? integer_max[cell] : integer_sums[cell], counts[cell]);
        }
        mine.groups.push_back(std::move(merged));
      }
      return;
    }

    // -- general path: every rank's table of records travels to every rank (one all-gather of the padded tables); merged in rank order,
    //    which is the order of first occurrence in the table because rank r holds the chunks before rank r + 1's
    const size_t record = 1 + 2 * n_keys + 3 * n_aggregates;   // first row, (key, is NULL)*, (value bits, integer, count)*   -- 8 bytes each
    const uint64_t my_groups = local.n;
    const auto group_counts = detail::all_gather_counts(comm, world, &my_groups, 1);
    const uint64_t most = *std::max_element(group_counts.begin(), group_counts.end());
    std::vector<int64_t> table(std::max<uint64_t>(1, most) * record, 0);
    for (uint32_t g = 0; g < local.n; ++g) {
      int64_t* cells = table.data() + size_t{g} * record;
      cells[0] = global_row(g);
      for (size_t k = 0; k < n_keys; ++k) { cells[1 + 2 * k] = local.key[k][g]; cells[2 + 2 * k] = local.key_null[k][g]; }
      for (size_t a = 0; a < n_aggregates; ++a) {
        std::memcpy(&cells[1 + 2 * n_keys + 3 * a], &local.value[a][g], sizeof(double));
        cells[2 + 2 * n_keys + 3 * a] = local.integer[a][g];
        cells[3 + 2 * n_keys + 3 * a] = local.count[a][g];
      }
    }
    const size_t piece = table.size() * sizeof(int64_t);
    DeviceBytes send(piece), recv(piece * world);
    check_status(hy_memcpy_h2d(send.get(), table.data(), piece));
    check_status(hy_comm_all_gather(comm, send.get(), recv.get(), piece));
    std::vector<int64_t> all(table.size() * world);
    check_status(hy_memcpy_d2h(all.data(), recv.get(), piece * world));
    struct Partial { double value = 0; int64_t integer = 0, count = 0; };
    std::map<std::vector<int64_t>, size_t> index;   // key cells (value, is NULL)* -> position in `merged_groups`
    std::vector<std::vector<Partial>> cells_of;
    for (uint32_t peer = 0; peer < world; ++peer) {
      for (uint64_t g = 0; g < group_counts[peer]; ++g) {
        const int64_t* cells = all.data() + (size_t{peer} * table.size()) + g * record;
        std::vector<int64_t> key(cells + 1, cells + 1 + 2 * n_keys);
        auto [slot, fresh] = index.emplace(key, mine.groups.size());
        if (fresh) {
          MergedGroup merged;
          merged.first_row = static_cast<uint64_t>(cells[0]);
          merged.key.resize(n_keys); merged.value.resize(n_aggregates); merged.integer.assign(n_aggregates, 0);
          for (size_t k = 0; k < n_keys; ++k) if (!key[2 * k + 1]) merged.key[k] = key[2 * k];
          mine.groups.push_back(std::move(merged));
          cells_of.emplace_back(n_aggregates);
        }
        auto& partial = cells_of[slot->second];
        for (size_t a = 0; a < n_aggregates; ++a) {
          double value;
          std::memcpy(&value, &cells[1 + 2 * n_keys + 3 * a], sizeof(double));
          const int64_t integer = cells[2 + 2 * n_keys + 3 * a], count = cells[3 + 2 * n_keys + 3 * a];
          if (!count && aggregates[a].function != WindowFunction::Count) continue;
          auto& p = partial[a];
          const bool seen = p.count != 0;
          if (aggregates[a].function == WindowFunction::Min) { p.value = seen ? std::min(p.value, value) : value; p.integer = seen ? std::min(p.integer, integer) : integer; }
          else if (aggregates[a].function == WindowFunction::Max) { p.value = seen ? std::max(p.value, value) : value; p.integer = seen ? std::max(p.integer, integer) : integer; }
          else { p.value += value; p.integer += integer; }
          p.count += count;
        }
      }
    }
    if (n_keys == 0 && mine.groups.empty()) {
      MergedGroup merged;
      merged.value.resize(n_aggregates); merged.integer.assign(n_aggregates, 0);
      mine.groups.push_back(std::move(merged));
      cells_of.emplace_back(n_aggregates);
    }
    for (size_t g = 0; g < mine.groups.size(); ++g)
      for (size_t a = 0; a < n_aggregates; ++a) finish(mine.groups[g], a, cells_of[g][a].value, cells_of[g][a].integer, cells_of[g][a].count);
  });
  for (uint32_t rank = 1; rank < world; ++rank) Assert(per_rank[rank].groups.size() == per_rank[0].groups.size(), "sharded_aggregate: the ranks disagree on the groups");
  return per_rank[0];
}

}  // namespace hyrise_amd
