// operator_chain.cpp -- TableScan -> JoinHash [-> AggregateHash] through `_on_execute()` of the C++ mirror (hyrise_amd/host/hyrise_host.hpp)
// at TPC-H SF10 size, with the intermediates as DevicePosLists in HBM.  What it checks and measures:
//   * the chain on device-resident intermediates against the SAME chain with host-memory results (device_resident_results() = false:
//     the boundary as rounds 1-5 used it) -- every PosList of every output table byte for byte, every aggregate row;
//   * with --oracle <liboracle.so>: against the CPU restatement of the reference's operators (oracle/, test infrastructure; loaded with
//     dlopen so that the timing runs bench.py makes of this binary never touch it) -- hyo_table_scan, hyo_join_hash over the scan's
//     PosLists as a reference column, hyo_aggregate_hash over the join's;
//   * --time N: milliseconds per chain, device-resident and host-result form side by side, as one JSON line (bench.py's
//     legs.cpp_operator_chain_ms).
// The plan (TPC-H shaped, tpch_queries.cpp Q3/Q4-like core): lineitem rows shipped before 1995-01-01 (ColumnVsValue on the dictionary-encoded
// l_shipdate, table_scan.cpp:97-240), joined with orders on the order key (JoinHash Inner, orders the build side, join_hash.cpp:116-225), grouped
// by l_returnflag with COUNT(*), SUM(l_quantity), MIN(l_shipdate) (AggregateHash, aggregate_hash.cpp:1180-1372).
// Usage: operator_chain [--orders N] [--oracle path] [--time N] [--calibrate K] [--threads T]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <random>

#include "../../hyrise_amd/host/hyrise_host.hpp"

using namespace hyrise_amd;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                                                      \
  do {                                                                                         \
    if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failures; } \
  } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- SF10-shaped tables (dbgen's shapes: sparse order keys -- 8 of every 32 --, one to seven lineitems per order, ship dates within 121 days
// of the order date; tpch_table_generator.cpp:155-166), encoded like Hyrise encodes them ---------------------------------
constexpr int32_t DAY_1995_01_01 = 1096;   // days since 1992-01-01

struct Tables {
  std::shared_ptr<Table> orders, lineitem;
};

template <typename Encode>
static void append_encoded(Segments& segments, std::vector<int32_t>& values, Encode encode) {
  ValueSegment<int32_t> plain(std::move(values), std::nullopt);
  segments.push_back(encode(plain));
  values.clear();
}

static Tables generate(uint64_t n_orders) {
  Tables t;
  t.orders = std::make_shared<Table>(TableColumnDefinitions{{"o_orderkey", DataType::Int, false}}, TableType::Data, Chunk::DEFAULT_SIZE);
  t.lineitem = std::make_shared<Table>(TableColumnDefinitions{{"l_orderkey", DataType::Int, false}, {"l_shipdate", DataType::Int, false}, {"l_returnflag", DataType::Int, false},
                                                              {"l_quantity", DataType::Int, false}}, TableType::Data, Chunk::DEFAULT_SIZE);
  std::mt19937_64 rng(42);
  std::vector<int32_t> keys, l_key, l_ship, l_flag, l_quantity;
  const auto flush_orders = [&] {
    if (keys.empty()) return;
    Segments segments;
    segments.push_back(std::make_shared<ValueSegment<int32_t>>(std::move(keys), std::nullopt));
    keys.clear();
    t.orders->append_chunk(std::move(segments));
  };
  const auto flush_lineitem = [&] {
    if (l_key.empty()) return;
    Segments segments;
    append_encoded(segments, l_key, [](const auto& s) { return ChunkEncoder::encode_frame_of_reference(s); });
    append_encoded(segments, l_ship, [](const auto& s) { return ChunkEncoder::encode_dictionary(s); });
    append_encoded(segments, l_flag, [](const auto& s) { return ChunkEncoder::encode_dictionary(s); });
    append_encoded(segments, l_quantity, [](const auto& s) { return ChunkEncoder::encode_dictionary(s); });
    t.lineitem->append_chunk(std::move(segments));
  };
  for (uint64_t i = 0; i < n_orders; ++i) {
    const int32_t key = static_cast<int32_t>((i / 8) * 32 + i % 8 + 1);
    keys.push_back(key);
    if (keys.size() == Chunk::DEFAULT_SIZE) flush_orders();
    const uint64_t r = rng();
    const int32_t order_date = static_cast<int32_t>(r % 2406);
    const uint32_t lines = 1 + static_cast<uint32_t>((r >> 16) % 7);
    for (uint32_t line = 0; line < lines; ++line) {
      const uint64_t q = rng();
      l_key.push_back(key);
      l_ship.push_back(order_date + 1 + static_cast<int32_t>(q % 121));
      l_flag.push_back(static_cast<int32_t>((q >> 8) % 3));
      l_quantity.push_back(1 + static_cast<int32_t>((q >> 16) % 50));
      if (l_key.size() == Chunk::DEFAULT_SIZE) flush_lineitem();
    }
  }
  flush_orders();
  flush_lineitem();
  return t;
}

static std::shared_ptr<TableWrapper> wrap(const std::shared_ptr<Table>& table) {
  auto wrapper = std::make_shared<TableWrapper>(table);
  wrapper->execute();
  return wrapper;
}

// ---- the chain ------------------------------------------------------------------------------------------------------------------------
struct ChainResult {
  std::shared_ptr<const Table> scanned, joined, aggregated;
  double scan_ms = 0, join_ms = 0, aggregate_ms = 0;
};

static ChainResult run_chain(const std::shared_ptr<TableWrapper>& orders, const std::shared_ptr<TableWrapper>& lineitem, bool with_aggregate) {
  ChainResult r;
  double t0 = now_ms();
  auto scan = std::make_shared<TableScan>(lineitem, ColumnID{1}, PredicateCondition::LessThan, AllTypeVariant{DAY_1995_01_01});
  scan->execute();
  r.scan_ms = now_ms() - t0;
  t0 = now_ms();
  auto join = std::make_shared<JoinHash>(orders, scan, JoinMode::Inner, ColumnIDPair{ColumnID{0}, ColumnID{0}});
  join->execute();
  r.join_ms = now_ms() - t0;
  r.scanned = scan->get_output();
  r.joined = join->get_output();
  if (with_aggregate) {
    t0 = now_ms();
    // columns of the join output: o_orderkey | l_orderkey, l_shipdate, l_returnflag, l_quantity
    auto aggregate = std::make_shared<AggregateHash>(join, std::vector<AggregateDefinition>{{INVALID_COLUMN_ID, WindowFunction::Count}, {ColumnID{4}, WindowFunction::Sum}, {ColumnID{2}, WindowFunction::Min}},
                                                     std::vector<ColumnID>{ColumnID{3}});
    aggregate->execute();
    r.aggregate_ms = now_ms() - t0;
    r.aggregated = aggregate->get_output();
  }
  return r;
}

// the RowIDs of one PosList, whatever its kind (DevicePosLists: one transfer per block, then views)
static std::vector<RowID> rows_of(const AbstractPosList& list, size_t block_rows_hint) {
  std::vector<RowID> rows;
  if (const auto* on_device = dynamic_cast<const DevicePosList*>(&list)) {
    if (block_rows_hint) on_device->block()->prefetch_to_host(block_rows_hint);
    const RowID* host = on_device->host_rows();
    rows.assign(host, host + list.size());
  } else if (const auto* plain = dynamic_cast<const RowIDPosList*>(&list)) {
    rows = plain->rows;
  } else {
    for (size_t i = 0; i < list.size(); ++i) rows.push_back(list[i]);
  }
  return rows;
}

static const AbstractPosList& pos_list_of(const std::shared_ptr<const Table>& table, ChunkID chunk, ColumnID column) {
  return *std::static_pointer_cast<ReferenceSegment>(table->get_chunk(chunk)->get_segment(column))->pos_list();
}

// two reference tables: same chunks, same PosLists (bytes), same guarantees; block_rows: rows per pooled block of `a`'s lists (0: copy list by list)
static void expect_same_reference_tables(const std::shared_ptr<const Table>& a, const std::shared_ptr<const Table>& b, const char* what, size_t block_rows) {
  EXPECT_TRUE(a->chunk_count() == b->chunk_count() && a->column_count() == b->column_count() && a->row_count() == b->row_count());
  if (a->chunk_count() != b->chunk_count() || a->column_count() != b->column_count()) return;
  uint64_t compared = 0;
  for (ChunkID chunk = 0; chunk < a->chunk_count(); ++chunk) {
    const AbstractPosList* last_a = nullptr;
    const AbstractPosList* last_b = nullptr;
    for (ColumnID column = 0; column < a->column_count(); ++column) {
      const auto& list_a = pos_list_of(a, chunk, column);
      const auto& list_b = pos_list_of(b, chunk, column);
      EXPECT_TRUE((&list_a == last_a) == (&list_b == last_b));   // columns share their PosLists in both or in neither
      if (&list_a == last_a && &list_b == last_b) continue;
      last_a = &list_a;
      last_b = &list_b;
      const auto rows_a = rows_of(list_a, block_rows), rows_b = rows_of(list_b, 0);
      const bool same = rows_a.size() == rows_b.size() && (rows_a.empty() || std::memcmp(rows_a.data(), rows_b.data(), rows_a.size() * sizeof(RowID)) == 0);
      if (!same) { std::printf("  %s: chunk %u column %u: PosLists differ (%zu vs %zu rows)\n", what, chunk, column, rows_a.size(), rows_b.size()); ++g_failures; return; }
      EXPECT_TRUE(list_a.references_single_chunk() == list_b.references_single_chunk());
      if (list_a.references_single_chunk() && list_a.size()) EXPECT_TRUE(list_a.common_chunk_id() == list_b.common_chunk_id());
      compared += rows_a.size();
    }
  }
  std::printf("  %s: %u chunks, %llu RowIDs byte-equal\n", what, a->chunk_count(), static_cast<unsigned long long>(compared));
}

static bool same_cell(const AllTypeVariant& a, const AllTypeVariant& b) {   // same alternative, same value (NULL == NULL)
  if (a.index() != b.index()) return false;
  return std::visit([&](const auto& x) -> bool {
    using T = std::decay_t<decltype(x)>;
    if constexpr (std::is_same_v<T, NullValue>) return true; else return x == std::get<T>(b);
  }, a);
}

static void expect_same_rows(const std::shared_ptr<const Table>& a, const std::shared_ptr<const Table>& b, const char* what) {
  const auto rows_a = a->get_rows(), rows_b = b->get_rows();
  EXPECT_TRUE(rows_a.size() == rows_b.size());
  for (size_t i = 0; i < std::min(rows_a.size(), rows_b.size()); ++i) {   // (group order included; integer aggregates: exact)
    EXPECT_TRUE(rows_a[i].size() == rows_b[i].size());
    for (size_t c = 0; c < std::min(rows_a[i].size(), rows_b[i].size()); ++c) EXPECT_TRUE(same_cell(rows_a[i][c], rows_b[i][c]));
  }
  std::printf("  %s: %zu rows equal\n", what, rows_a.size());
}

// ---- the oracle (dlopen: test runs only) ---------------------------------------------------------------------------------------------------
struct OracleColumn {   // hyo_column of oracle/hy_oracle.h
  const hy_segment* segments;
  uint32_t n_chunks;
};
struct Oracle {
  int32_t (*table_scan)(const OracleColumn*, const hy_predicate*, hy_scan_result*, int) = nullptr;
  int32_t (*join_hash)(const OracleColumn*, const OracleColumn*, uint32_t, hy_join_result*, int) = nullptr;
  int32_t (*aggregate_hash)(const OracleColumn* const*, uint32_t, const uint32_t*, const OracleColumn* const*, uint32_t, hy_aggregate_result*) = nullptr;
  explicit Oracle(const char* path) {
    void* handle = dlopen(path, RTLD_NOW);
    if (!handle) Fail(std::string("cannot load the oracle: ") + dlerror());
    table_scan = reinterpret_cast<decltype(table_scan)>(dlsym(handle, "hyo_table_scan"));
    join_hash = reinterpret_cast<decltype(join_hash)>(dlsym(handle, "hyo_join_hash"));
    aggregate_hash = reinterpret_cast<decltype(aggregate_hash)>(dlsym(handle, "hyo_aggregate_hash"));
    if (!table_scan || !join_hash || !aggregate_hash) Fail("the oracle lacks an entry point");
  }
};

// a reference table's column as the oracle reads it: one HY_ENC_REFERENCE descriptor per chunk over host PosLists
struct OracleReferenceColumn {
  std::vector<std::vector<RowID>> lists;
  std::vector<hy_segment> segments;
  OracleColumn column{};
  void add(std::vector<RowID> rows, const OracleColumn* referenced, ChunkID common_chunk) {   // common_chunk: 0xFFFFFFFF = no single-chunk guarantee
    lists.push_back(std::move(rows));
    hy_segment d{};
    d.encoding = HY_ENC_REFERENCE; d.data_type = HY_TYPE_INT; d.width = 8;
    d.size = static_cast<uint32_t>(lists.back().size());
    d.ref = reinterpret_cast<const hy_column*>(referenced);
    d.ref_chunk_id = common_chunk;
    segments.push_back(d);
  }
  void finish(const std::vector<bool>& entire) {
    for (size_t i = 0; i < segments.size(); ++i) segments[i].data = entire[i] ? nullptr : lists[i].data();
    column.segments = segments.data();
    column.n_chunks = static_cast<uint32_t>(segments.size());
  }
};

static void check_against_oracle(const Oracle& oracle, const Tables& tables, const ChainResult& got, int threads) {
  const auto l_orderkey = device_column(tables.lineitem, ColumnID{0}), l_shipdate = device_column(tables.lineitem, ColumnID{1}), l_returnflag = device_column(tables.lineitem, ColumnID{2}),
             l_quantity = device_column(tables.lineitem, ColumnID{3}), o_orderkey = device_column(tables.orders, ColumnID{0});
  const auto as_oracle = [](const std::shared_ptr<DeviceColumn>& c) { return OracleColumn{c->descriptors.data(), static_cast<uint32_t>(c->descriptors.size())}; };
  const OracleColumn key_l = as_oracle(l_orderkey), ship = as_oracle(l_shipdate), flag = as_oracle(l_returnflag), quantity = as_oracle(l_quantity), key_o = as_oracle(o_orderkey);
  const auto n_chunks = tables.lineitem->chunk_count();
  const uint64_t rows = tables.lineitem->row_count();

  // TableScan
  std::vector<RowID> matches(rows);
  std::vector<uint64_t> offsets(n_chunks + 1);
  std::vector<uint32_t> counts(n_chunks);
  std::vector<uint8_t> states(n_chunks);
  hy_scan_result scan{};
  scan.mem = HY_MEM_HOST; scan.matches = reinterpret_cast<hy_row_id*>(matches.data()); scan.capacity = rows;
  scan.offsets = offsets.data(); scan.counts = counts.data(); scan.chunk_state = states.data();
  hy_predicate predicate{};
  predicate.condition = HY_PRED_LESS_THAN; predicate.value_type = HY_TYPE_INT; predicate.value.i32 = DAY_1995_01_01;
  EXPECT_TRUE(oracle.table_scan(&ship, &predicate, &scan, threads) == HY_OK);
  OracleReferenceColumn scanned_keys;   // the scan's output as the join's probe column: l_orderkey through the PosLists
  std::vector<bool> entire;
  ChunkID out_chunk = 0;
  uint64_t scan_rows = 0;
  for (ChunkID c = 0; c < n_chunks; ++c) {
    if (!counts[c]) continue;
    EXPECT_TRUE(out_chunk < got.scanned->chunk_count());
    if (out_chunk >= got.scanned->chunk_count()) return;
    const auto& list = pos_list_of(got.scanned, out_chunk, ColumnID{0});
    const bool all = counts[c] == tables.lineitem->get_chunk(c)->size();
    std::vector<RowID> want;
    if (all) for (uint32_t i = 0; i < counts[c]; ++i) want.push_back(RowID{c, i});   // (the oracle writes nothing for an ALL_MATCH chunk it elides)
    else want.assign(matches.begin() + offsets[c], matches.begin() + offsets[c] + counts[c]);
    const auto have = rows_of(list, rows);
    if (have.size() != want.size() || std::memcmp(have.data(), want.data(), want.size() * sizeof(RowID)) != 0) { std::printf("  scan: chunk %u differs from the oracle\n", c); ++g_failures; return; }
    EXPECT_TRUE((dynamic_cast<const EntireChunkPosList*>(&list) != nullptr) == all);
    scanned_keys.add(std::move(want), &key_l, c);
    entire.push_back(all);
    scan_rows += counts[c];
    ++out_chunk;
  }
  EXPECT_TRUE(out_chunk == got.scanned->chunk_count());
  scanned_keys.finish(entire);
  std::printf("  TableScan vs oracle: %llu RowIDs in %u output chunks byte-equal\n", static_cast<unsigned long long>(scan_rows), out_chunk);

  // JoinHash(orders, scan output)
  const uint64_t capacity = std::max<uint64_t>(tables.orders->row_count(), scan_rows);
  const uint32_t slice_capacity = static_cast<uint32_t>(capacity / 131070 + n_chunks + 1000);
  std::vector<RowID> left(capacity), right(capacity);
  std::vector<uint64_t> slice_offsets(slice_capacity + 2);
  hy_join_result join{};
  join.mem = HY_MEM_HOST; join.radix_bits = 0xFFFFFFFFu; join.left_pos = reinterpret_cast<hy_row_id*>(left.data()); join.right_pos = reinterpret_cast<hy_row_id*>(right.data());
  join.capacity = capacity; join.slice_offsets = slice_offsets.data(); join.slice_capacity = slice_capacity;
  EXPECT_TRUE(oracle.join_hash(&key_o, &scanned_keys.column, HY_JOIN_INNER, &join, threads) == HY_OK);
  EXPECT_TRUE(join.n_pairs == got.joined->row_count());
  std::vector<uint64_t> chunk_offsets(join.n_slices + 1);
  uint32_t n_out = 0;
  check_status(hy_join_output_chunks(slice_offsets.data(), join.n_slices, chunk_offsets.data(), &n_out));
  EXPECT_TRUE(n_out == got.joined->chunk_count());
  if (n_out != got.joined->chunk_count()) return;
  // the right side's positions name rows of the scan's output: dereferenced through its PosLists (join_output_writing.cpp:95-200)
  std::vector<RowID> right_rows(join.n_pairs);
  for (uint64_t i = 0; i < join.n_pairs; ++i) right_rows[i] = scanned_keys.lists[right[i].chunk_id][right[i].chunk_offset];
  OracleReferenceColumn joined_flag, joined_quantity, joined_ship;
  std::vector<bool> none;
  for (uint32_t k = 0; k < n_out; ++k) {
    const uint64_t begin = chunk_offsets[k], end = chunk_offsets[k + 1];
    const auto have_left = rows_of(pos_list_of(got.joined, k, ColumnID{0}), join.n_pairs), have_right = rows_of(pos_list_of(got.joined, k, ColumnID{1}), join.n_pairs);
    const bool same = have_left.size() == end - begin && have_right.size() == end - begin && std::memcmp(have_left.data(), left.data() + begin, (end - begin) * sizeof(RowID)) == 0 &&
                      std::memcmp(have_right.data(), right_rows.data() + begin, (end - begin) * sizeof(RowID)) == 0;
    if (!same) { std::printf("  join: output chunk %u differs from the oracle\n", k); ++g_failures; return; }
    std::vector<RowID> chunk_rows(right_rows.begin() + begin, right_rows.begin() + end);
    joined_flag.add(chunk_rows, &flag, 0xFFFFFFFFu);
    joined_quantity.add(chunk_rows, &quantity, 0xFFFFFFFFu);
    joined_ship.add(std::move(chunk_rows), &ship, 0xFFFFFFFFu);
    none.push_back(false);
  }
  joined_flag.finish(none); joined_quantity.finish(none); joined_ship.finish(none);
  std::printf("  JoinHash vs oracle: %llu pairs in %u output chunks, both PosLists byte-equal (radix bits %u)\n", static_cast<unsigned long long>(join.n_pairs), n_out, join.radix_bits);

  // AggregateHash over the join's output
  if (!got.aggregated) return;
  const uint32_t group_capacity = 64;
  std::vector<RowID> group_rows(group_capacity);
  std::vector<std::vector<uint64_t>> values(3, std::vector<uint64_t>(group_capacity));
  std::vector<std::vector<uint8_t>> nulls(3, std::vector<uint8_t>(group_capacity));
  std::vector<hy_aggregate_column> columns(3);
  for (int a = 0; a < 3; ++a) { columns[a].values = values[a].data(); columns[a].is_null = nulls[a].data(); }
  hy_aggregate_result aggregate{};
  aggregate.mem = HY_MEM_HOST; aggregate.group_capacity = group_capacity; aggregate.group_row_ids = reinterpret_cast<hy_row_id*>(group_rows.data()); aggregate.columns = columns.data();
  const OracleColumn* groupby[] = {&joined_flag.column};
  const uint32_t functions[] = {HY_AGG_COUNT, HY_AGG_SUM, HY_AGG_MIN};
  const OracleColumn* inputs[] = {nullptr, &joined_quantity.column, &joined_ship.column};
  EXPECT_TRUE(oracle.aggregate_hash(groupby, 1, functions, inputs, 3, &aggregate) == HY_OK);
  const auto have = got.aggregated->get_rows();
  EXPECT_TRUE(have.size() == aggregate.n_groups);
  for (uint32_t g = 0; g < std::min<size_t>(aggregate.n_groups, have.size()); ++g) {
    // the representative row is a row of the JOIN OUTPUT (chunk k, offset o): its l_returnflag through the right side's PosList
    const RowID at = group_rows[g];
    const RowID base = joined_flag.lists[at.chunk_id][at.chunk_offset];
    const auto want_flag = (*tables.lineitem->get_chunk(base.chunk_id)->get_segment(ColumnID{2}))[base.chunk_offset];
    EXPECT_TRUE(same_cell(have[g][0], want_flag));
    EXPECT_TRUE(std::get<int64_t>(have[g][1]) == static_cast<int64_t>(values[0][g]));
    EXPECT_TRUE(std::get<int64_t>(have[g][2]) == static_cast<int64_t>(values[1][g]));
    EXPECT_TRUE(std::get<int32_t>(have[g][3]) == reinterpret_cast<const int32_t*>(values[2].data())[g]);
  }
  std::printf("  AggregateHash vs oracle: %u groups, keys / COUNT / SUM / MIN equal, same order\n", aggregate.n_groups);
}

template <typename Run>
static double median_ms(int runs, Run run) {
  std::vector<double> times;
  for (int i = 0; i < runs; ++i) { const double t0 = now_ms(); run(); times.push_back(now_ms() - t0); }
  std::sort(times.begin(), times.end());
  return times[times.size() / 2];
}

int main(int argc, char** argv) {
  uint64_t n_orders = 15000000;
  const char* oracle_path = nullptr;
  int time_runs = 0, threads = 8;
  uint32_t calibrate = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string arg = argv[i];
    if (arg == "--orders" && i + 1 < argc) n_orders = std::strtoull(argv[++i], nullptr, 10);
    else if (arg == "--oracle" && i + 1 < argc) oracle_path = argv[++i];
    else if (arg == "--time" && i + 1 < argc) time_runs = std::atoi(argv[++i]);
    else if (arg == "--threads" && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (arg == "--calibrate" && i + 1 < argc) calibrate = static_cast<uint32_t>(std::atoi(argv[++i]));
    else { std::fprintf(stderr, "usage: operator_chain [--orders N] [--oracle liboracle.so] [--time N] [--calibrate K] [--threads T]\n"); return 2; }
  }
  try {
    check_status(hy_init(0));
    double t0 = now_ms();
    const Tables tables = generate(n_orders);
    std::printf("tables: orders %llu rows / %u chunks, lineitem %llu rows / %u chunks (%.1f s)\n", static_cast<unsigned long long>(tables.orders->row_count()), tables.orders->chunk_count(),
                static_cast<unsigned long long>(tables.lineitem->row_count()), tables.lineitem->chunk_count(), (now_ms() - t0) / 1e3);
    const auto orders = wrap(tables.orders), lineitem = wrap(tables.lineitem);
    for (ColumnID c = 0; c < 4; ++c) (void)device_column(tables.lineitem, c);   // (residency: the columns are uploaded once, as in Hyrise's buffer of encoded segments)
    (void)device_column(tables.orders, ColumnID{0});
    std::vector<float> calibration(calibrate);
    uint32_t chosen = 0;
    if (calibrate) {   // the library's result-buffer pool, calibrated for this process's largest join
      check_status(hy_result_pool_calibrate(device_column(tables.orders, ColumnID{0})->handle, device_column(tables.lineitem, ColumnID{0})->handle, HY_JOIN_INNER, tables.lineitem->row_count(),
                                            calibrate, 0, calibration.data(), &chosen));
    }

    device_resident_results() = true;
    const ChainResult on_device = run_chain(orders, lineitem, true);
    std::printf("device-resident chain: scan %.3f ms, join %.3f ms, aggregate %.3f ms (first run)\n", on_device.scan_ms, on_device.join_ms, on_device.aggregate_ms);
    EXPECT_TRUE(on_device.scanned->chunk_count() > 0 && dynamic_cast<const DevicePosList*>(&pos_list_of(on_device.joined, 0, ColumnID{0})) != nullptr);
    if (!time_runs || oracle_path) {
      device_resident_results() = false;
      const ChainResult on_host = run_chain(orders, lineitem, true);
      device_resident_results() = true;
      std::printf("host-result chain: scan %.3f ms, join %.3f ms, aggregate %.3f ms\n", on_host.scan_ms, on_host.join_ms, on_host.aggregate_ms);
      expect_same_reference_tables(on_device.scanned, on_host.scanned, "TableScan output, device-resident vs host-result", tables.lineitem->row_count());
      expect_same_reference_tables(on_device.joined, on_host.joined, "JoinHash output, device-resident vs host-result", on_device.joined->row_count());
      expect_same_rows(on_device.aggregated, on_host.aggregated, "AggregateHash output, device-resident vs host-result");
    }
    if (oracle_path) check_against_oracle(Oracle(oracle_path), tables, on_device, threads);

    if (time_runs) {
      const auto chain = [&](bool device, bool aggregate) { device_resident_results() = device; (void)run_chain(orders, lineitem, aggregate); device_resident_results() = true; };
      const auto join_only = [&](bool device) {
        device_resident_results() = device;
        auto join = std::make_shared<JoinHash>(orders, lineitem, JoinMode::Inner, ColumnIDPair{ColumnID{0}, ColumnID{0}});
        join->execute();
        device_resident_results() = true;
      };
      const auto scan_only = [&](bool device) {
        device_resident_results() = device;
        auto scan = std::make_shared<TableScan>(lineitem, ColumnID{1}, PredicateCondition::LessThan, AllTypeVariant{DAY_1995_01_01});
        scan->execute();
        device_resident_results() = true;
      };
      for (int i = 0; i < 2; ++i) { chain(true, true); join_only(true); scan_only(true); }
      const double scan_join_device = median_ms(time_runs, [&] { chain(true, false); }), chain_device = median_ms(time_runs, [&] { chain(true, true); });
      const double join_device = median_ms(time_runs, [&] { join_only(true); }), scan_device = median_ms(time_runs, [&] { scan_only(true); });
      const int host_runs = std::max(1, std::min(time_runs, 3));
      const double scan_join_host = median_ms(host_runs, [&] { chain(false, false); }), join_host = median_ms(host_runs, [&] { join_only(false); }), scan_host = median_ms(host_runs, [&] { scan_only(false); });
      std::printf("{\"cpp_operator_chain_ms\": {\"scan_join_aggregate_device_resident\": %.4f, \"scan_join_device_resident\": %.4f, \"scan_join_host_result\": %.4f, "
                  "\"join_orders_lineitem_device_resident\": %.4f, \"join_orders_lineitem_host_result\": %.4f, \"scan_device_resident\": %.4f, \"scan_host_result\": %.4f, "
                  "\"rows\": {\"orders\": %llu, \"lineitem\": %llu, \"scan_matches\": %llu, \"join_pairs\": %llu}, \"pool_candidates\": %u, \"pool_chosen\": %u}}\n",
                  chain_device, scan_join_device, scan_join_host, join_device, join_host, scan_device, scan_host, static_cast<unsigned long long>(tables.orders->row_count()),
                  static_cast<unsigned long long>(tables.lineitem->row_count()), static_cast<unsigned long long>(on_device.scanned->row_count()),
                  static_cast<unsigned long long>(on_device.joined->row_count()), calibrate, chosen);
    }
  } catch (const std::exception& e) {
    std::printf("EXCEPTION: %s\n", e.what());
    ++g_failures;
  }
  hy_shutdown();
  std::printf("%s\n", g_failures ? "OPERATOR CHAIN FAILED" : "OPERATOR CHAIN OK");
  return g_failures ? 1 : 0;
}
