// multi_gpu_tests.cpp -- the C++ coordinator of the sharded operators (hyrise_amd/host/multi_gpu.hpp) through the real RCCL calls:
// one process, one worker thread and one communicator per device (ncclCommInitAll).  Every sharded result is compared with the
// single-GPU operator of hyrise_host.hpp on the whole table (which tests/cpp/host_tests.cpp holds against the reference's fixtures)
// and, for the joins, with a nested loop on the host.
// Usage: multi_gpu_tests [device ...]   (default: every device of the box; "0 0" asks RCCL for two ranks on one GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <set>

#include "../../hyrise_amd/host/multi_gpu.hpp"

using namespace hyrise_amd;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                                                      \
  do {                                                                                         \
    if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failures; } \
  } while (0)

static void run(const char* name, const std::function<void()>& test) {
  const int before = g_failures;
  try { test(); } catch (const std::exception& e) { std::printf("  EXCEPTION: %s\n", e.what()); ++g_failures; }
  std::printf("[%s] %s\n", g_failures == before ? "  OK  " : "FAILED", name);
  std::fflush(stdout);
}

static uint64_t g_state = 88172645463325252ull;
static uint64_t next_random() { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return g_state; }

static std::shared_ptr<TableWrapper> wrap(std::shared_ptr<Table> table) {
  auto wrapper = std::make_shared<TableWrapper>(std::move(table));
  wrapper->execute();
  return wrapper;
}

// returnflag-like and linestatus-like keys, a wide key, an int and a double measure (both with NULLs)
static std::shared_ptr<Table> measures_table(size_t rows, ChunkOffset chunk_rows) {
  TableColumnDefinitions definitions{{"small_a", DataType::Int, false}, {"small_b", DataType::Long, true}, {"wide", DataType::Int, false},
                                     {"quantity", DataType::Int, true}, {"price", DataType::Double, true}};
  auto table = std::make_shared<Table>(definitions, TableType::Data, chunk_rows);
  for (size_t r = 0; r < rows; ++r) {
    std::vector<AllTypeVariant> row;
    row.emplace_back(static_cast<int32_t>(60 + next_random() % 3));
    if (next_random() % 17 == 0) row.emplace_back(NullValue{}); else row.emplace_back(static_cast<int64_t>(next_random() % 2) - 1);
    row.emplace_back(static_cast<int32_t>(next_random() % 100000) * 7 - 1000);
    if (next_random() % 11 == 0) row.emplace_back(NullValue{}); else row.emplace_back(static_cast<int32_t>(next_random() % 50) - 5);
    if (next_random() % 13 == 0) row.emplace_back(NullValue{}); else row.emplace_back(static_cast<double>(next_random() % 100000) / 100.0);
    table->append(std::move(row));
  }
  table->finalize();
  return table;
}

static void check_aggregate(DeviceGroup& group, const std::shared_ptr<Table>& table, const std::vector<ColumnID>& groupby, bool allow_all_reduce, bool expect_all_reduce) {
  const std::vector<AggregateDefinition> definitions{{ColumnID{3}, WindowFunction::Sum}, {ColumnID{4}, WindowFunction::Sum}, {ColumnID{4}, WindowFunction::Avg},
                                                     {ColumnID{3}, WindowFunction::Min}, {ColumnID{4}, WindowFunction::Max}, {ColumnID{3}, WindowFunction::Count},
                                                     {INVALID_COLUMN_ID, WindowFunction::Count}, {ColumnID{3}, WindowFunction::Avg}};
  auto single = std::make_shared<AggregateHash>(wrap(table), definitions, groupby);
  single->execute();
  const auto expected = single->get_output()->get_rows();

  std::map<ColumnID, ShardedColumn> columns;
  for (const ColumnID id : {ColumnID{0}, ColumnID{1}, ColumnID{2}, ColumnID{3}, ColumnID{4}}) columns[id] = shard_column(group, table, id);
  std::vector<const ShardedColumn*> keys;
  for (const auto id : groupby) keys.push_back(&columns[id]);
  std::vector<ShardedAggregate> aggregates;
  for (const auto& d : definitions) aggregates.push_back({d.function, d.column_id == INVALID_COLUMN_ID ? nullptr : &columns[d.column_id]});
  const auto merged = sharded_aggregate(group, keys, aggregates, allow_all_reduce);
  EXPECT_TRUE(merged.used_all_reduce == expect_all_reduce);
  EXPECT_TRUE(merged.groups.size() == expected.size());
  if (merged.groups.size() != expected.size()) { std::printf("  %zu groups, expected %zu\n", merged.groups.size(), expected.size()); return; }
  for (size_t g = 0; g < expected.size(); ++g) {   // the same groups in the same order
    const auto& row = expected[g];
    const auto& got = merged.groups[g];
    bool same = true;
    for (size_t k = 0; k < groupby.size(); ++k) {
      if (variant_is_null(row[k])) { same &= !got.key[k].has_value(); continue; }
      const int64_t key = row[k].index() == 1 ? std::get<int32_t>(row[k]) : std::get<int64_t>(row[k]);
      same &= got.key[k].has_value() && *got.key[k] == key;
    }
    for (size_t a = 0; a < definitions.size(); ++a) {
      const auto& cell = row[groupby.size() + a];
      if (variant_is_null(cell)) { same &= !got.value[a].has_value(); continue; }
      if (!got.value[a].has_value()) { same = false; continue; }
      if (cell.index() == 1) same &= got.integer[a] == std::get<int32_t>(cell);
      else if (cell.index() == 2) same &= got.integer[a] == std::get<int64_t>(cell);   // integer sums, counts, minima: exact
      else {
        const double want = cell.index() == 3 ? std::get<float>(cell) : std::get<double>(cell);
        same &= std::fabs(*got.value[a] - want) <= 1e-9 * std::max(1.0, std::fabs(want));   // double sums in another order
      }
    }
    if (!same) { std::printf("  group %zu differs\n", g); ++g_failures; return; }
  }
}

static std::shared_ptr<Table> key_table(size_t rows, ChunkOffset chunk_rows, uint64_t key_range, bool with_nulls, bool unique, DataType type) {
  TableColumnDefinitions definitions{{"key", type, with_nulls}};
  auto table = std::make_shared<Table>(definitions, TableType::Data, chunk_rows);
  for (size_t r = 0; r < rows; ++r) {
    std::vector<AllTypeVariant> row;
    const int64_t key = unique ? static_cast<int64_t>(r * 3 + 1) : static_cast<int64_t>(next_random() % key_range) * 3 + 1;
    if (with_nulls && next_random() % 19 == 0) row.emplace_back(NullValue{});
    else if (type == DataType::Int) row.emplace_back(static_cast<int32_t>(key));
    else row.emplace_back(key);
    table->append(std::move(row));
  }
  table->finalize();
  return table;
}

using Pairs = std::multiset<std::pair<uint64_t, uint64_t>>;
static uint64_t packed(const RowID& row) { return uint64_t{row.chunk_id} << 32 | row.chunk_offset; }

static std::vector<std::pair<RowID, std::optional<int64_t>>> keyed_rows(const std::shared_ptr<Table>& table) {
  std::vector<std::pair<RowID, std::optional<int64_t>>> out;
  for (ChunkID c = 0; c < table->chunk_count(); ++c) {
    const auto segment = table->get_chunk(c)->get_segment(ColumnID{0});
    for (ChunkOffset o = 0; o < segment->size(); ++o) {
      const auto value = (*segment)[o];
      std::optional<int64_t> key;
      if (!variant_is_null(value)) key = value.index() == 1 ? std::get<int32_t>(value) : std::get<int64_t>(value);
      out.push_back({RowID{c, o}, key});
    }
  }
  return out;
}

static Pairs nested_loop(const std::shared_ptr<Table>& left, const std::shared_ptr<Table>& right, JoinMode mode) {
  std::multimap<int64_t, RowID> build;
  for (const auto& [row, key] : keyed_rows(right)) if (key) build.insert({*key, row});
  Pairs pairs;
  for (const auto& [row, key] : keyed_rows(left)) {
    const auto [begin, end] = key ? build.equal_range(*key) : std::make_pair(build.end(), build.end());
    if (mode == JoinMode::Inner || mode == JoinMode::Left) {
      for (auto it = begin; it != end; ++it) pairs.insert({packed(row), packed(it->second)});
      if (mode == JoinMode::Left && begin == end) pairs.insert({packed(row), packed(NULL_ROW_ID)});
    } else if ((mode == JoinMode::Semi) == (begin != end)) {
      pairs.insert({packed(row), 0});
    }
  }
  return pairs;
}

static Pairs collected(const ShardedJoinOutput& out, bool with_right, bool swap_sides = false) {
  Pairs pairs;
  for (size_t rank = 0; rank < out.left.size(); ++rank) {
    for (size_t i = 0; i < out.left[rank].size(); ++i) {
      const uint64_t l = packed(out.left[rank][i]), r = with_right ? packed(out.right[rank][i]) : 0;
      pairs.insert(swap_sides ? std::make_pair(r, l) : std::make_pair(l, r));
    }
  }
  return pairs;
}

int main(int argc, char** argv) {
  int32_t count = 0;
  check_status(hy_device_count(&count));
  std::vector<int32_t> devices;
  for (int i = 1; i < argc; ++i) devices.push_back(std::atoi(argv[i]));
  if (devices.empty()) for (int32_t d = 0; d < count; ++d) devices.push_back(d);
  check_status(hy_init(devices[0]));
  std::printf("devices:");
  for (const auto d : devices) std::printf(" %d", d);
  std::printf("  (world %zu)\n", devices.size());
  std::fflush(stdout);
  std::unique_ptr<DeviceGroup> group;
  try {
    group = std::make_unique<DeviceGroup>(devices);
  } catch (const std::exception& e) {
    std::printf("COMMUNICATOR NOT CREATED: %s\n", e.what());
    return 3;
  }

  run("collectives: all_reduce, all_gather, all_to_all_v on every worker", [&] {
    const uint32_t world = group->size();
    std::vector<int> ok(world, 0);
    group->run([&](uint32_t rank, hy_comm* comm) {
      uint32_t r = 99, w = 0;
      check_status(hy_comm_rank(comm, &r, &w));
      bool good = r == rank && w == world;
      std::vector<int64_t> cells{int64_t{rank} + 1, -int64_t{rank}}, sums(2), minima(2);
      DeviceBytes buffer(16), out(16);
      check_status(hy_memcpy_h2d(buffer.get(), cells.data(), 16));
      check_status(hy_comm_all_reduce(comm, buffer.get(), out.get(), 2, HY_TYPE_LONG, HY_COMM_SUM));
      check_status(hy_memcpy_d2h(sums.data(), out.get(), 16));
      check_status(hy_comm_all_reduce(comm, buffer.get(), out.get(), 2, HY_TYPE_LONG, HY_COMM_MIN));
      check_status(hy_memcpy_d2h(minima.data(), out.get(), 16));
      good &= sums[0] == int64_t{world} * (world + 1) / 2 && sums[1] == -int64_t{world} * (world - 1) / 2 && minima[0] == 1 && minima[1] == -int64_t{world - 1};
      std::vector<double> reals{0.5 * (rank + 1)}, real_sum(1);
      check_status(hy_memcpy_h2d(buffer.get(), reals.data(), 8));
      check_status(hy_comm_all_reduce(comm, buffer.get(), out.get(), 1, HY_TYPE_DOUBLE, HY_COMM_MAX));
      check_status(hy_memcpy_d2h(real_sum.data(), out.get(), 8));
      good &= real_sum[0] == 0.5 * world;
      // all_to_all_v: rank r sends (p + 1) values r * 100 + p to peer p
      std::vector<uint64_t> send_bytes(world), recv_bytes(world);
      std::vector<int32_t> send;
      for (uint32_t p = 0; p < world; ++p) { send_bytes[p] = (p + 1) * 4; recv_bytes[p] = (rank + 1) * 4; for (uint32_t i = 0; i <= p; ++i) send.push_back(rank * 100 + p); }
      DeviceBytes send_device(send.size() * 4), recv_device(size_t{world} * (rank + 1) * 4);
      check_status(hy_memcpy_h2d(send_device.get(), send.data(), send.size() * 4));
      check_status(hy_comm_all_to_all_v(comm, send_device.get(), send_bytes.data(), recv_device.get(), recv_bytes.data()));
      std::vector<int32_t> received(size_t{world} * (rank + 1));
      check_status(hy_memcpy_d2h(received.data(), recv_device.get(), received.size() * 4));
      for (uint32_t p = 0; p < world; ++p) for (uint32_t i = 0; i <= rank; ++i) good &= received[p * (rank + 1) + i] == static_cast<int32_t>(p * 100 + rank);
      ok[rank] = good;
    });
    for (const auto good : ok) EXPECT_TRUE(good);
  });

  run("a column of another device is refused", [&] {
    if (group->size() < 2 || group->device(0) == group->device(1)) return;
    const auto table = key_table(100, 50, 10, false, false, DataType::Int);
    const auto sharded = shard_column(*group, table, ColumnID{0});
    bool refused = false;
    group->run([&](uint32_t rank, hy_comm*) {
      if (rank != 0) return;
      uint64_t counts[2];
      refused = hy_repartition_count(sharded.shard[1]->handle, 2, counts) == HY_ERR_INVALID;
    });
    EXPECT_TRUE(refused);
  });

  const auto measures = measures_table(23'456, 1000);
  run("sharded AggregateHash: fixed slots + all-reduce (two small keys, NULL group)", [&] { check_aggregate(*group, measures, {ColumnID{0}, ColumnID{1}}, true, true); });
  run("sharded AggregateHash: the same through the all-gather of (key, partial) tables", [&] { check_aggregate(*group, measures, {ColumnID{0}, ColumnID{1}}, false, false); });
  run("sharded AggregateHash: a wide key (all-gather path)", [&] { check_aggregate(*group, measures, {ColumnID{2}, ColumnID{0}}, true, false); });
  run("sharded AggregateHash: no GROUP BY", [&] { check_aggregate(*group, measures, {}, true, true); });

  for (const auto type : {DataType::Int, DataType::Long}) {
    const auto left = key_table(40'000, 4096, 9000, true, false, type), right = key_table(9'500, 1024, 9000, true, false, type);
    const auto left_sharded = shard_column(*group, left, ColumnID{0}), right_sharded = shard_column(*group, right, ColumnID{0});
    const std::string name = type == DataType::Int ? "int" : "long";
    run(("hash repartition join, Inner, " + name + " keys with duplicates and NULLs").c_str(), [&] {
      EXPECT_TRUE(collected(sharded_join_repartition(*group, left_sharded, right_sharded, JoinMode::Inner), true) == nested_loop(left, right, JoinMode::Inner));
    });
    run(("hash repartition join, Semi, " + name + " keys").c_str(), [&] {
      EXPECT_TRUE(collected(sharded_join_repartition(*group, left_sharded, right_sharded, JoinMode::Semi), false) == nested_loop(left, right, JoinMode::Semi));
    });
  }
  {
    const auto dimension = key_table(8'000, 1000, 0, false, true, DataType::Int), fact = key_table(50'000, 4096, 9000, true, false, DataType::Int);
    const auto dimension_sharded = shard_column(*group, dimension, ColumnID{0}), fact_sharded = shard_column(*group, fact, ColumnID{0});
    run("broadcast-build join, Inner, build = left (primary keys x foreign keys)", [&] {
      EXPECT_TRUE(collected(sharded_join_broadcast(*group, dimension_sharded, fact_sharded, JoinMode::Inner, true), true) == nested_loop(dimension, fact, JoinMode::Inner));
    });
    run("broadcast-build join, Inner / Left / Semi / AntiNullAsFalse, build = right", [&] {
      EXPECT_TRUE(collected(sharded_join_broadcast(*group, dimension_sharded, fact_sharded, JoinMode::Inner, false), true) == nested_loop(fact, dimension, JoinMode::Inner));
      EXPECT_TRUE(collected(sharded_join_broadcast(*group, dimension_sharded, fact_sharded, JoinMode::Left, false), true) == nested_loop(fact, dimension, JoinMode::Left));
      EXPECT_TRUE(collected(sharded_join_broadcast(*group, dimension_sharded, fact_sharded, JoinMode::Semi, false), false) == nested_loop(fact, dimension, JoinMode::Semi));
      EXPECT_TRUE(collected(sharded_join_broadcast(*group, dimension_sharded, fact_sharded, JoinMode::AntiNullAsFalse, false), false) == nested_loop(fact, dimension, JoinMode::AntiNullAsFalse));
    });
    run("broadcast-build join refuses a gathered probe side", [&] {
      bool threw = false;
      try { sharded_join_broadcast(*group, dimension_sharded, fact_sharded, JoinMode::Semi, true); } catch (const std::logic_error&) { threw = true; }
      EXPECT_TRUE(threw);
    });
  }
  group.reset();
  hy_shutdown();
  std::printf("%s\n", g_failures ? "MULTI GPU TESTS FAILED" : "MULTI GPU TESTS PASSED");
  return g_failures ? 1 : 0;
}
