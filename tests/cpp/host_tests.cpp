// host_tests.cpp -- the reference's operator tests, re-stated against the C++ mirror of the operator interface
// (hyrise_amd/host/hyrise_host.hpp) and therefore running on the HIP library through the C ABI.  Sources mirrored:
//   src/test/lib/operators/table_scan_test.cpp   (ScanOnCompressedSegments :407-431, ScanOnReferencedCompressedSegments
//                                                 :433-463, out-of-range literals :486-534, SingleScan :296-302,
//                                                 DoubleScan, ScanForNullValues :661-684, string scans)
//   src/test/lib/operators/join_test_runner.cpp  (differential test against a nested-loop join, :656-791)
//   src/test/lib/operators/aggregate_test.cpp    (input/expected .tbl pairs, EXPECT_TABLE_EQ_UNORDERED)
// Usage: host_tests <tests/golden/tbl directory>.  Prints one line per test, exits non-zero on the first failure.
#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>
#include <map>
#include <random>
#include <array>

#include "../../hyrise_amd/host/hyrise_host.hpp"

using namespace hyrise_amd;

static std::string g_tbl;
static int g_failures = 0;

#define EXPECT_TRUE(cond)                                                                      \
  do {                                                                                         \
    if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failures; } \
  } while (0)

static bool cells_equal(const AllTypeVariant& a, const AllTypeVariant& b) {   // check_table_equal.cpp:34,109-115 (lenient types)
  if (variant_is_null(a) || variant_is_null(b)) return variant_is_null(a) && variant_is_null(b);
  if (a.index() == 5 || b.index() == 5) return a.index() == b.index() && std::get<std::string>(a) == std::get<std::string>(b);
  const auto as_double = [](const AllTypeVariant& v) {
    switch (v.index()) { case 1: return static_cast<double>(std::get<int32_t>(v)); case 2: return static_cast<double>(std::get<int64_t>(v));
                         case 3: return static_cast<double>(std::get<float>(v)); default: return std::get<double>(v); }
  };
  const double x = as_double(a), y = as_double(b);
  return std::fabs(x - y) < std::max(1e-4, std::fabs(y) * 1e-4);
}

static bool tables_equal_unordered(const std::shared_ptr<const Table>& got, const std::shared_ptr<const Table>& want) {
  auto a = got->get_rows(), b = want->get_rows();
  if (a.size() != b.size()) { std::printf("  row counts differ: %zu vs %zu\n", a.size(), b.size()); return false; }
  for (const auto& row : a) {
    bool matched = false;
    for (size_t i = 0; i < b.size() && !matched; ++i) {
      if (b[i].size() != row.size()) continue;
      bool equal = true;
      for (size_t c = 0; c < row.size() && equal; ++c) equal = cells_equal(row[c], b[i][c]);
      if (equal) { b.erase(b.begin() + i); matched = true; }
    }
    if (!matched) return false;
  }
  return true;
}

static std::vector<int32_t> column_ints(const std::shared_ptr<const Table>& table, ColumnID column) {
  std::vector<int32_t> out;
  for (const auto& row : table->get_rows()) out.push_back(std::get<int32_t>(row[column]));
  std::sort(out.begin(), out.end());
  return out;
}

static std::shared_ptr<TableWrapper> wrap(std::shared_ptr<Table> table) {
  auto wrapper = std::make_shared<TableWrapper>(std::move(table));
  wrapper->execute();
  return wrapper;
}

static std::shared_ptr<TableWrapper> load_and_encode(const std::string& file, ChunkOffset chunk_size, EncodingType encoding) {
  auto table = load_table(g_tbl + "/" + file, chunk_size);
  ChunkEncoder::encode_all_chunks(table, encoding);
  return wrap(table);
}

static void run(const char* name, const std::function<void()>& test) {
  const int before = g_failures;
  try { test(); } catch (const std::exception& e) { std::printf("  EXCEPTION: %s\n", e.what()); ++g_failures; }
  std::printf("[%s] %s\n", g_failures == before ? "  OK  " : "FAILED", name);
}

static const EncodingType ENCODINGS[] = {EncodingType::Unencoded, EncodingType::Dictionary, EncodingType::FrameOfReference, EncodingType::LZ4};

static void test_scan_on_compressed_segments() {   // table_scan_test.cpp:407-431, 486-534
  using PC = PredicateCondition;
  const std::vector<int32_t> all{100, 100, 102, 102, 104, 104, 106, 106, 108, 108, 110, 110, 112, 112};
  struct Case { int32_t literal; std::map<PC, std::vector<int32_t>> expected; };
  const std::vector<Case> cases = {
      {6, {{PC::Equals, {106, 106}}, {PC::NotEquals, {100, 100, 102, 102, 104, 104, 108, 108, 110, 110, 112, 112}}, {PC::LessThan, {100, 100, 102, 102, 104, 104}},
           {PC::LessThanEquals, {100, 100, 102, 102, 104, 104, 106, 106}}, {PC::GreaterThan, {108, 108, 110, 110, 112, 112}},
           {PC::GreaterThanEquals, {106, 106, 108, 108, 110, 110, 112, 112}}, {PC::IsNull, {}}, {PC::IsNotNull, all}}},
      {30, {{PC::Equals, {}}, {PC::NotEquals, all}, {PC::LessThan, all}, {PC::LessThanEquals, all}, {PC::GreaterThan, {}}, {PC::GreaterThanEquals, {}}}},
      {-10, {{PC::Equals, {}}, {PC::NotEquals, all}, {PC::LessThan, {}}, {PC::LessThanEquals, {}}, {PC::GreaterThan, all}, {PC::GreaterThanEquals, all}}}};
  for (const auto encoding : ENCODINGS) {
    for (const auto& [file, chunk] : std::vector<std::pair<std::string, ChunkOffset>>{{"int_int_shuffled.tbl", 7}, {"int_int_shuffled_2.tbl", 5}}) {
      const auto wrapper = load_and_encode(file, chunk, encoding);
      for (const auto& c : cases) {
        for (const auto& [condition, expected] : c.expected) {
          auto scan = std::make_shared<TableScan>(wrapper, ColumnID{0}, condition, AllTypeVariant{c.literal});
          scan->execute();
          EXPECT_TRUE(column_ints(scan->get_output(), ColumnID{1}) == expected);
        }
      }
    }
  }
}

static void test_scan_on_referenced_segments() {   // table_scan_test.cpp:433-463
  using PC = PredicateCondition;
  const std::map<PC, std::vector<int32_t>> expected = {{PC::Equals, {104, 104}}, {PC::NotEquals, {100, 100, 102, 102, 106, 106}}, {PC::LessThan, {100, 100, 102, 102}},
      {PC::LessThanEquals, {100, 100, 102, 102, 104, 104}}, {PC::GreaterThan, {106, 106}}, {PC::GreaterThanEquals, {104, 104, 106, 106}}, {PC::IsNull, {}},
      {PC::IsNotNull, {100, 100, 102, 102, 104, 104, 106, 106}}};
  for (const auto encoding : ENCODINGS) {
    const auto wrapper = load_and_encode("int_int_shuffled.tbl", 7, encoding);
    for (const auto& [condition, values] : expected) {
      auto scan1 = std::make_shared<TableScan>(wrapper, ColumnID{1}, PC::LessThan, AllTypeVariant{int32_t{108}});
      scan1->execute();
      auto scan2 = std::make_shared<TableScan>(scan1, ColumnID{0}, condition, AllTypeVariant{int32_t{4}});
      scan2->execute();
      EXPECT_TRUE(column_ints(scan2->get_output(), ColumnID{1}) == values);
    }
  }
}

static void test_single_and_double_scan() {   // table_scan_test.cpp:296-302 and DoubleScan
  for (const auto encoding : ENCODINGS) {
    const auto wrapper = load_and_encode("int_float.tbl", 2, encoding);
    auto scan = std::make_shared<TableScan>(wrapper, ColumnID{0}, PredicateCondition::GreaterThanEquals, AllTypeVariant{int32_t{1234}});
    scan->execute();
    EXPECT_TRUE(tables_equal_unordered(scan->get_output(), load_table(g_tbl + "/int_float_filtered2.tbl", 1)));
    auto scan2 = std::make_shared<TableScan>(scan, ColumnID{1}, PredicateCondition::LessThan, AllTypeVariant{457.9f});
    scan2->execute();
    EXPECT_TRUE(tables_equal_unordered(scan2->get_output(), load_table(g_tbl + "/int_float_filtered.tbl", 1)));
    auto between = std::make_shared<TableScan>(wrapper, ColumnID{0}, PredicateCondition::BetweenInclusive, AllTypeVariant{int32_t{1234}}, AllTypeVariant{int32_t{20000}});
    between->execute();
    EXPECT_TRUE(tables_equal_unordered(between->get_output(), load_table(g_tbl + "/int_float_filtered2.tbl", 1)));
  }
}

static void test_scan_for_null_values() {   // table_scan_test.cpp:661-684
  for (const auto encoding : ENCODINGS) {
    const auto wrapper = load_and_encode("int_int_w_null_8_rows.tbl", 4, encoding);
    auto is_null = std::make_shared<TableScan>(wrapper, ColumnID{1}, PredicateCondition::IsNull);
    is_null->execute();
    EXPECT_TRUE(column_ints(is_null->get_output(), ColumnID{0}) == (std::vector<int32_t>{12, 123}));
    auto not_null = std::make_shared<TableScan>(wrapper, ColumnID{1}, PredicateCondition::IsNotNull);
    not_null->execute();
    EXPECT_TRUE(not_null->get_output()->row_count() == 6);
  }
}

static void test_string_dictionary_scan() {   // the l_shipdate case: DictionarySegment<pmr_string>, value ids resolved per chunk
  const auto wrapper = load_and_encode("int_string_like.tbl", 2, EncodingType::Dictionary);
  const auto rows = wrapper->get_output()->get_rows();
  for (const auto condition : {PredicateCondition::Equals, PredicateCondition::NotEquals, PredicateCondition::LessThan, PredicateCondition::GreaterThanEquals}) {
    const std::string literal = std::get<std::string>(rows[2][1]);
    auto scan = std::make_shared<TableScan>(wrapper, ColumnID{1}, condition, AllTypeVariant{literal});
    scan->execute();
    size_t expected = 0;
    for (const auto& row : rows) {
      if (variant_is_null(row[1])) continue;   // NULL never matches a comparison
      const auto& v = std::get<std::string>(row[1]);
      expected += condition == PredicateCondition::Equals ? v == literal : condition == PredicateCondition::NotEquals ? v != literal
                : condition == PredicateCondition::LessThan ? v < literal : v >= literal;
    }
    EXPECT_TRUE(scan->get_output()->row_count() == expected);
  }
}

static void test_like_on_dictionary_segments() {   // table_scan_string_test.cpp:112-327 (the *OnDictSegment / OnReferencedDictSegment cases)
  struct Case { PredicateCondition condition; const char* pattern; const char* expected; };
  const Case cases[] = {{PredicateCondition::Like, "%", "int_string_like_without_null.tbl"}, {PredicateCondition::Like, "Dampf%", "int_string_like_starting.tbl"},
                        {PredicateCondition::Like, "%gesellschaft", "int_string_like_ending.tbl"}, {PredicateCondition::Like, "%schifffahrtsgesellschaft%", "int_string_like_containing.tbl"},
                        {PredicateCondition::Like, "Schiff%schaft", "int_string_like_containing_wildcard.tbl"}, {PredicateCondition::Like, "%D%_m_f%", "int_string_like_starting.tbl"},
                        {PredicateCondition::Like, "%not_there%", nullptr}, {PredicateCondition::NotLike, "%", nullptr},
                        {PredicateCondition::NotLike, "%foo%", "int_string_like_without_null.tbl"}, {PredicateCondition::NotLike, "D_m_f%", "int_string_like_not_starting.tbl"},
                        {PredicateCondition::LikeInsensitive, "dampf%", "int_string_like_starting.tbl"}};
  const auto wrapper = load_and_encode("int_string_like.tbl", 5, EncodingType::Dictionary);
  for (const auto& c : cases) {
    for (const bool referenced : {false, true}) {
      std::shared_ptr<const AbstractOperator> input = wrapper;
      if (referenced) {
        auto first = std::make_shared<TableScan>(wrapper, ColumnID{0}, PredicateCondition::GreaterThan, AllTypeVariant{int32_t{0}});
        first->execute();
        input = first;
      }
      auto scan = std::make_shared<TableScan>(input, ColumnID{1}, c.condition, AllTypeVariant{std::string(c.pattern)});
      scan->execute();
      if (!c.expected) { EXPECT_TRUE(scan->get_output()->row_count() == 0); if (!referenced) EXPECT_TRUE(scan->num_chunks_with_early_out == 2); continue; }
      EXPECT_TRUE(tables_equal_unordered(scan->get_output(), load_table(g_tbl + "/" + c.expected, 1)));
    }
  }
  // special characters are literals (:193-219)
  const auto special = load_and_encode("int_string_like_special_chars.tbl", 2, EncodingType::Dictionary);
  const std::pair<const char*, const char*> special_cases[] = {{"%2^2%", "int_string_like_special_chars_1.tbl"}, {"%$%$%", "int_string_like_special_chars_1.tbl"},
      {"%(%)%", "int_string_like_special_chars_2.tbl"}, {"%la\\.^$+?)({}.*__bl%", "int_string_like_special_chars_3.tbl"}};
  for (const auto& [pattern, expected] : special_cases) {
    auto scan = std::make_shared<TableScan>(special, ColumnID{1}, PredicateCondition::Like, AllTypeVariant{std::string(pattern)});
    scan->execute();
    EXPECT_TRUE(tables_equal_unordered(scan->get_output(), load_table(g_tbl + "/" + expected, 1)));
  }
  // LIKE on a non-string column / with a non-string pattern throws (:91-101)
  bool thrown = false;
  try { auto s = std::make_shared<TableScan>(load_and_encode("int_float.tbl", 2, EncodingType::Dictionary), ColumnID{0}, PredicateCondition::Like, AllTypeVariant{std::string("%test")}); s->execute(); }
  catch (const std::logic_error&) { thrown = true; }
  EXPECT_TRUE(thrown);
  thrown = false;
  try { auto s = std::make_shared<TableScan>(wrapper, ColumnID{1}, PredicateCondition::Like, AllTypeVariant{int32_t{1234}}); s->execute(); }
  catch (const std::logic_error&) { thrown = true; }
  EXPECT_TRUE(thrown);
}

static void test_type_mismatch_throws() {   // table_scan_test.cpp:383-405 (EXPECT_THROW std::logic_error)
  const auto wrapper = load_and_encode("int_float.tbl", 2, EncodingType::Unencoded);
  auto scan = std::make_shared<TableScan>(wrapper, ColumnID{0}, PredicateCondition::Equals, AllTypeVariant{std::string("x")});
  bool thrown = false;
  try { scan->execute(); } catch (const std::logic_error&) { thrown = true; }
  EXPECT_TRUE(thrown);
}

static void test_literals_of_other_types_are_cast_without_loss() {   // table_scan.cpp:336-366, 406-448; lossless_predicate_cast_test.cpp
  for (const auto encoding : {EncodingType::Unencoded, EncodingType::Dictionary}) {
    const auto wrapper = load_and_encode("int_float.tbl", 2, encoding);   // b: 458.7f, 456.7f, 457.7f
    const auto count = [&](ColumnID column, PredicateCondition condition, AllTypeVariant value, std::optional<AllTypeVariant> value2 = std::nullopt) {
      auto scan = std::make_shared<TableScan>(wrapper, column, condition, std::move(value), std::move(value2));
      scan->execute();
      return scan->get_output()->row_count();
    };
    // a double literal against the float column: 457.7 (double) lies just below 457.7f, so < and <= both exclude that row,
    // > and >= both include it
    EXPECT_TRUE(static_cast<double>(457.7f) > 457.7);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::LessThan, 457.7) == 1);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::LessThanEquals, 457.7) == 1);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::GreaterThan, 457.7) == 2);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::GreaterThanEquals, 457.7) == 2);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::GreaterThanEquals, static_cast<double>(457.7f)) == 2);   // exactly a float: no adjustment
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::GreaterThan, static_cast<double>(457.7f)) == 1);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::BetweenExclusive, 456.7, 458.0) == 2);                   // (456.7, 458): 456.7f > 456.7 is inside
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::BetweenInclusive, int32_t{457}, int64_t{458}) == 1);
    // long / double literals against the int column
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::GreaterThan, int64_t{1000}) == 2);
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::Equals, 123.0) == 1);
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::Equals, 123.0f) == 1);
    // No lossless cast: the reference runs its ExpressionEvaluator scan (table_scan.cpp:346-366, 450) -- the adapter the stock operator,
    // this mirror its row-by-row stand-in; queries that run on stock Hyrise run through the adapter (a: 12345, 123, 1234).
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::LessThan, 123.5) == 1);
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::GreaterThanEquals, 123.5) == 2);
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::Equals, 123.5) == 0);
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::LessThan, int64_t{100'000'000'000}) == 3);
    EXPECT_TRUE(count(ColumnID{0}, PredicateCondition::BetweenInclusive, 122.5, 1234.5) == 2);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::Equals, 457.7) == 0);      // float_column = a double that is no float: no row
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::NotEquals, 457.7) == 3);
    EXPECT_TRUE(count(ColumnID{1}, PredicateCondition::Equals, 3.1) == 0);
  }
}

static void test_scan_project_aggregate() {   // the fused pass behind the operator interface: Q6's shape, with and without Validate
  // a: 0 .. 99, b = a % 7 as float, c = 0.5 * a as double, key = a % 3; chunks of 32 rows
  auto table = std::make_shared<Table>(TableColumnDefinitions{{"a", DataType::Int, false}, {"b", DataType::Float, false}, {"c", DataType::Double, true}, {"key", DataType::Int, false}},
                                       TableType::Data, ChunkOffset{32});
  for (ChunkID chunk = 0; chunk * 32 < 100; ++chunk) {
    std::vector<int32_t> a, key;
    std::vector<float> b;
    std::vector<double> c;
    std::vector<bool> c_null;
    for (int32_t i = static_cast<int32_t>(chunk) * 32; i < std::min<int32_t>(100, (static_cast<int32_t>(chunk) + 1) * 32); ++i) {
      a.push_back(i); key.push_back(i % 3); b.push_back(static_cast<float>(i % 7)); c.push_back(0.5 * i); c_null.push_back(i % 10 == 9);
    }
    const auto rows = a.size();
    table->append_chunk({std::make_shared<ValueSegment<int32_t>>(a, std::nullopt), std::make_shared<ValueSegment<float>>(b, std::nullopt),
                         std::make_shared<ValueSegment<double>>(c, c_null), std::make_shared<ValueSegment<int32_t>>(key, std::nullopt)}, std::make_shared<MvccData>(rows));
  }
  // SELECT key, SUM(b * (1 - c)), COUNT(*), MIN(a) FROM t WHERE a BETWEEN 10 AND 89 AND b < 5 GROUP BY key    (rows with c NULL: NULL product, not summed)
  const auto expected = [&](const std::function<bool(int32_t)>& visible) {
    std::map<int32_t, std::tuple<double, int64_t, int32_t, bool>> groups;   // sum, count, min, any non-NULL product
    std::vector<int32_t> order;
    for (int32_t i = 0; i < 100; ++i) {
      if (!visible(i) || i < 10 || i > 89 || static_cast<float>(i % 7) >= 5.0f) continue;
      if (!groups.count(i % 3)) { groups[i % 3] = {0.0, 0, i, false}; order.push_back(i % 3); }
      auto& [sum, count, minimum, any] = groups[i % 3];
      if (i % 10 != 9) { sum += static_cast<double>(static_cast<float>(i % 7)) * (1.0 - 0.5 * i); any = true; }
      count += 1;
      minimum = std::min(minimum, i);
    }
    return std::make_pair(groups, order);
  };
  const std::vector<ScanPredicate> predicates = {{ColumnID{0}, PredicateCondition::BetweenInclusive, AllTypeVariant{int32_t{10}}, AllTypeVariant{int64_t{89}}},
                                                 {ColumnID{1}, PredicateCondition::LessThan, AllTypeVariant{5.0}, std::nullopt}};
  const std::vector<ExpressionAggregate> aggregates = {
      {WindowFunction::Sum, {ExpressionNode::column(ColumnID{1}), ExpressionNode::literal(int32_t{1}), ExpressionNode::column(ColumnID{2}), ExpressionNode::arithmetic(ArithmeticOperator::Subtraction),
                             ExpressionNode::arithmetic(ArithmeticOperator::Multiplication)}},
      {WindowFunction::Count, {}}, {WindowFunction::Min, {ExpressionNode::column(ColumnID{0})}}};
  const auto check = [&](const std::shared_ptr<const Table>& out, const std::function<bool(int32_t)>& visible) {
    const auto [groups, order] = expected(visible);
    EXPECT_TRUE(out->row_count() == groups.size());
    int32_t previous_key = -1;
    for (uint64_t g = 0; g < out->row_count() && g < order.size(); ++g) {   // one int32 GROUP BY column: the immediate-key shortcut, key order (aggregate_hash.cpp:770-804)
      const auto row = out->get_rows()[g];
      const int32_t key = std::get<int32_t>(row[0]);
      EXPECT_TRUE(groups.count(key) == 1 && key > previous_key);
      previous_key = key;
      if (!groups.count(key)) continue;
      const auto& [sum, count, minimum, any] = groups.at(key);
      EXPECT_TRUE(any ? std::abs(std::get<double>(row[1]) - sum) <= 1e-9 * std::max(1.0, std::abs(sum)) : variant_is_null(row[1]));
      EXPECT_TRUE(std::get<int64_t>(row[2]) == count);
      EXPECT_TRUE(std::get<int32_t>(row[3]) == minimum);
    }
  };
  auto fused = std::make_shared<ScanProjectAggregate>(wrap(table), predicates, std::vector<ColumnID>{ColumnID{3}}, aggregates);
  fused->execute();
  check(fused->get_output(), [](int32_t) { return true; });
  // behind Validate: rows 20 .. 39 were deleted before the snapshot, rows 50 .. 54 are another transaction's uncommitted inserts
  for (int32_t i = 0; i < 100; ++i) {
    const auto& mvcc = table->get_chunk(static_cast<ChunkID>(i / 32))->mvcc_data();
    mvcc->set_begin_cid(static_cast<ChunkOffset>(i % 32), 1);
    if (i >= 20 && i < 40) mvcc->set_end_cid(static_cast<ChunkOffset>(i % 32), 3);
    if (i >= 50 && i < 55) { mvcc->set_begin_cid(static_cast<ChunkOffset>(i % 32), MAX_COMMIT_ID); mvcc->set_tid(static_cast<ChunkOffset>(i % 32), 9); }
  }
  auto validated = std::make_shared<ScanProjectAggregate>(wrap(table), predicates, std::vector<ColumnID>{ColumnID{3}}, aggregates);
  validated->set_transaction_context(std::make_shared<TransactionContext>(7, 5));
  validated->execute();
  check(validated->get_output(), [](int32_t i) { return !(i >= 20 && i < 40) && !(i >= 50 && i < 55); });
  // ... and equal to the chain Validate -> TableScan -> TableScan on the survivors' count
  auto validate = std::make_shared<Validate>(wrap(table));
  validate->set_transaction_context(std::make_shared<TransactionContext>(7, 5));
  validate->execute();
  auto first = std::make_shared<TableScan>(validate, ColumnID{0}, PredicateCondition::BetweenInclusive, AllTypeVariant{int32_t{10}}, AllTypeVariant{int32_t{89}});
  first->execute();
  auto second = std::make_shared<TableScan>(first, ColumnID{1}, PredicateCondition::LessThan, AllTypeVariant{5.0f});
  second->execute();
  int64_t counted = 0;
  for (const auto& row : validated->get_output()->get_rows()) counted += std::get<int64_t>(row[2]);
  EXPECT_TRUE(static_cast<uint64_t>(counted) == second->get_output()->row_count());
}

static void test_star_join_aggregate() {   // SSB's plan shape as one operator, against nested loops over the same tables
  std::mt19937 rng(17);
  const auto make = [](std::vector<std::string> names, ChunkOffset chunk) {
    TableColumnDefinitions definitions;
    for (auto& name : names) definitions.push_back({name, DataType::Int, false});
    return std::make_shared<Table>(definitions, TableType::Data, chunk);
  };
  auto part = make({"p_key", "p_brand", "p_category"}, 40);        // filtered dimension, sparse keys
  auto date = make({"d_key", "d_year"}, 16);                       // unfiltered dimension
  auto fact = make({"f_part", "f_date", "f_revenue", "f_cost"}, 700);
  std::vector<std::array<int32_t, 3>> parts;
  for (int32_t i = 0; i < 300; ++i) { parts.push_back({i * 7 + 3, static_cast<int32_t>(rng() % 9), static_cast<int32_t>(rng() % 5)}); part->append({parts.back()[0], parts.back()[1], parts.back()[2]}); }
  std::vector<std::array<int32_t, 2>> dates;
  for (int32_t i = 0; i < 50; ++i) { dates.push_back({19920101 + i, 1992 + i / 10}); date->append({dates.back()[0], dates.back()[1]}); }
  struct Cell { int64_t revenue = 0, profit = 0, count = 0; };
  std::map<std::pair<int32_t, int32_t>, Cell> expected;
  uint64_t joined = 0;
  for (int32_t i = 0; i < 5000; ++i) {
    const bool dangling = rng() % 10 == 0;
    const int32_t p = dangling ? 1 : parts[rng() % parts.size()][0], d = dates[rng() % dates.size()][0];
    const int32_t revenue = static_cast<int32_t>(rng() % 10000), cost = static_cast<int32_t>(rng() % 6000);
    fact->append({p, d, revenue, cost});
    for (const auto& row : parts) {
      if (row[0] != p || row[2] != 2) continue;   // p_category = 2
      auto& cell = expected[std::make_pair(dates[static_cast<size_t>(d - 19920101)][1], row[1])];
      cell.revenue += revenue; cell.profit += revenue - cost; ++cell.count; ++joined;
    }
  }
  part->finalize(); date->finalize(); fact->finalize();
  for (const auto encoding : {EncodingType::Unencoded, EncodingType::FrameOfReference, EncodingType::Dictionary}) {
    ChunkEncoder::encode_all_chunks(fact, encoding);
    ChunkEncoder::encode_all_chunks(part, encoding);
    const std::vector<StarDimension> dimensions = {{wrap(part), ColumnID{0}, ScanPredicate{ColumnID{2}, PredicateCondition::Equals, AllTypeVariant{int64_t{2}}, std::nullopt}, ColumnID{0}},
                                                   {wrap(date), ColumnID{0}, std::nullopt, ColumnID{1}}};
    auto star = std::make_shared<StarJoinAggregate>(wrap(fact), dimensions, std::vector<StarColumn>{{2, ColumnID{1}}, {1, ColumnID{1}}},
                                                    std::vector<StarAggregate>{{WindowFunction::Sum, StarColumn{0, ColumnID{2}}, std::nullopt, std::nullopt},
                                                                               {WindowFunction::Sum, StarColumn{0, ColumnID{2}}, ArithmeticOperator::Subtraction, StarColumn{0, ColumnID{3}}},
                                                                               {WindowFunction::Count, std::nullopt, std::nullopt, std::nullopt}});
    star->execute();
    const auto out = star->get_output();
    EXPECT_TRUE(star->joined_rows == joined && out->row_count() == expected.size());
    for (const auto& row : out->get_rows()) {
      const auto it = expected.find(std::make_pair(std::get<int32_t>(row[0]), std::get<int32_t>(row[1])));
      EXPECT_TRUE(it != expected.end());
      if (it == expected.end()) continue;
      EXPECT_TRUE(std::get<int64_t>(row[2]) == it->second.revenue && std::get<int64_t>(row[3]) == it->second.profit && std::get<int64_t>(row[4]) == it->second.count);
    }
  }
}

static void test_join_output_chunks_are_merged() {   // join_output_writing.cpp:245-296: PosLists below 1000 rows merge up to 4000
  const auto left = load_and_encode("join_test_runner/input_table_left_15.tbl", 3, EncodingType::Unencoded);
  const auto right = load_and_encode("join_test_runner/input_table_right_10.tbl", 3, EncodingType::Unencoded);
  auto join = std::make_shared<JoinHash>(left, right, JoinMode::Inner, ColumnIDPair{ColumnID{0}, ColumnID{0}}, 4);   // 16 radix partitions
  join->execute();
  EXPECT_TRUE(join->radix_bits == 4);
  EXPECT_TRUE(join->get_output()->row_count() > 0 && join->get_output()->chunk_count() == 1);   // a handful of rows: one merged chunk
  // a result of several thousand rows: chunks of 1000 .. 3999 rows, none smaller except the last
  auto big_left = std::make_shared<Table>(TableColumnDefinitions{{"a", DataType::Int, false}}, TableType::Data, ChunkOffset{1000});
  auto big_right = std::make_shared<Table>(TableColumnDefinitions{{"a", DataType::Int, false}}, TableType::Data, ChunkOffset{1000});
  for (int32_t i = 0; i < 6000; ++i) { big_left->append({AllTypeVariant{i}}); big_right->append({AllTypeVariant{(i * 7) % 6000}}); }
  big_left->finalize();
  big_right->finalize();
  auto big = std::make_shared<JoinHash>(wrap(big_left), wrap(big_right), JoinMode::Inner, ColumnIDPair{ColumnID{0}, ColumnID{0}}, 5);   // 32 partitions of ~188 rows
  big->execute();
  const auto out = big->get_output();
  EXPECT_TRUE(out->row_count() == 6000 && out->chunk_count() < 7 && out->chunk_count() >= 2);
  for (ChunkID c = 0; c + 1 < out->chunk_count(); ++c) EXPECT_TRUE(out->get_chunk(c)->size() >= 1000 && out->get_chunk(c)->size() < 4000);
  for (const auto& row : out->get_rows()) EXPECT_TRUE(cells_equal(row[0], row[1]));
}

static void test_join_against_nested_loop() {   // join_test_runner.cpp:656-791 (Inner / Semi / AntiNullAsFalse on int columns)
  for (const auto chunk : {ChunkOffset{10}, ChunkOffset{3}}) {
    for (const auto encoding : {EncodingType::Unencoded, EncodingType::Dictionary}) {
      const auto left = load_and_encode("join_test_runner/input_table_left_15.tbl", chunk, encoding);
      const auto right = load_and_encode("join_test_runner/input_table_right_10.tbl", chunk, encoding);
      const auto lrows = left->get_output()->get_rows(), rrows = right->get_output()->get_rows();
      for (const ColumnID column : {ColumnID{0}, ColumnID{1}}) {   // int, int_null
        size_t inner = 0, semi = 0, anti = 0;
        for (const auto& l : lrows) {
          bool any = false;
          for (const auto& r : rrows) {
            if (!variant_is_null(l[column]) && !variant_is_null(r[column]) && std::get<int32_t>(l[column]) == std::get<int32_t>(r[column])) { ++inner; any = true; }
          }
          semi += any;
          anti += !any;
        }
        auto join = std::make_shared<JoinHash>(left, right, JoinMode::Inner, ColumnIDPair{column, column});
        join->execute();
        EXPECT_TRUE(join->get_output()->row_count() == inner);
        for (const auto& row : join->get_output()->get_rows()) EXPECT_TRUE(cells_equal(row[column], row[lrows[0].size() + column]));
        auto semi_join = std::make_shared<JoinHash>(left, right, JoinMode::Semi, ColumnIDPair{column, column}, 2);
        semi_join->execute();
        EXPECT_TRUE(semi_join->get_output()->row_count() == semi);
        auto anti_join = std::make_shared<JoinHash>(left, right, JoinMode::AntiNullAsFalse, ColumnIDPair{column, column});
        anti_join->execute();
        EXPECT_TRUE(anti_join->get_output()->row_count() == anti);
        auto left_join = std::make_shared<JoinHash>(left, right, JoinMode::Left, ColumnIDPair{column, column});
        left_join->execute();
        EXPECT_TRUE(left_join->get_output()->row_count() == inner + anti);
      }
    }
  }
}

static bool cxx_equal(const AllTypeVariant& x, const AllTypeVariant& y) {   // x == y as C++ compares the two alternatives' types
  return std::visit([](const auto& a, const auto& b) -> bool {
    using A = std::decay_t<decltype(a)>;
    using B = std::decay_t<decltype(b)>;
    if constexpr (std::is_arithmetic_v<A> && std::is_arithmetic_v<B>) return a == b;
    else return false;
  }, x, y);
}

static void test_join_on_mixed_numeric_key_types() {   // join_test_runner.cpp:183-193: every data type against every other one
  const auto left = load_and_encode("join_test_runner/input_table_left_15.tbl", 4, EncodingType::Dictionary);
  const auto right = load_and_encode("join_test_runner/input_table_right_10.tbl", 3, EncodingType::Unencoded);
  const auto lrows = left->get_output()->get_rows(), rrows = right->get_output()->get_rows();
  for (ColumnID lc{0}; lc < 8; ++lc) {     // int, int_null, float, float_null, double, double_null, long, long_null
    for (ColumnID rc{0}; rc < 8; ++rc) {
      size_t inner = 0, anti = 0;
      for (const auto& l : lrows) {
        bool any = false;
        for (const auto& r : rrows) {
          if (cxx_equal(l[lc], r[rc])) { ++inner; any = true; }
        }
        anti += !any;
      }
      auto join = std::make_shared<JoinHash>(left, right, JoinMode::Inner, ColumnIDPair{lc, rc});
      join->execute();
      EXPECT_TRUE(join->get_output()->row_count() == inner);
      for (const auto& row : join->get_output()->get_rows()) EXPECT_TRUE(cxx_equal(row[lc], row[lrows[0].size() + rc]));
      auto left_join = std::make_shared<JoinHash>(left, right, JoinMode::Left, ColumnIDPair{lc, rc}, 2);
      left_join->execute();
      EXPECT_TRUE(left_join->get_output()->row_count() == inner + anti);
    }
  }
}

static void test_join_on_string_keys() {   // join_test_runner.cpp:183-193 (string x string); ids from string_join_id
  for (const auto chunk : {ChunkOffset{4}, ChunkOffset{10}}) {
    const auto left = load_and_encode("join_test_runner/input_table_left_15.tbl", chunk, EncodingType::Dictionary);
    const auto right = load_and_encode("join_test_runner/input_table_right_10.tbl", 3, EncodingType::Dictionary);
    const auto lrows = left->get_output()->get_rows(), rrows = right->get_output()->get_rows();
    for (const ColumnID column : {ColumnID{8}, ColumnID{9}}) {   // string, string_null
      size_t inner = 0, semi = 0;
      for (const auto& l : lrows) {
        bool any = false;
        for (const auto& r : rrows) {
          if (!variant_is_null(l[column]) && !variant_is_null(r[column]) && std::get<std::string>(l[column]) == std::get<std::string>(r[column])) { ++inner; any = true; }
        }
        semi += any;
      }
      auto join = std::make_shared<JoinHash>(left, right, JoinMode::Inner, ColumnIDPair{column, column}, 3);
      join->execute();
      EXPECT_TRUE(inner > 0 && join->get_output()->row_count() == inner);
      for (const auto& row : join->get_output()->get_rows()) EXPECT_TRUE(cells_equal(row[column], row[lrows[0].size() + column]));
      auto semi_join = std::make_shared<JoinHash>(left, right, JoinMode::Semi, ColumnIDPair{column, column});
      semi_join->execute();
      EXPECT_TRUE(semi_join->get_output()->row_count() == semi);
      auto right_join = std::make_shared<JoinHash>(left, right, JoinMode::Right, ColumnIDPair{column, column});
      right_join->execute();
      size_t unmatched_right = 0;
      for (const auto& r : rrows) {
        bool any = false;
        for (const auto& l : lrows) any = any || (!variant_is_null(l[column]) && !variant_is_null(r[column]) && std::get<std::string>(l[column]) == std::get<std::string>(r[column]));
        unmatched_right += !any;
      }
      EXPECT_TRUE(right_join->get_output()->row_count() == inner + unmatched_right);
    }
  }
  // libstdc++'s std::hash<std::string>, spelled out -- and this binary is built with libstdc++
  for (const std::string s : {"", "a", "abcdefgh", "Dampfschifffahrtsgesellschaft"}) EXPECT_TRUE(libstdcxx_hash_bytes(s.data(), s.size()) == std::hash<std::string>{}(s));
}

static void test_join_with_secondary_predicates() {   // join_test_runner.cpp:207-211, 464-480: {0,0} <, >=, != as secondary predicates
  const auto left = load_and_encode("join_test_runner/input_table_left_15.tbl", 3, EncodingType::Dictionary);
  const auto right = load_and_encode("join_test_runner/input_table_right_10.tbl", 10, EncodingType::Unencoded);
  const auto lrows = left->get_output()->get_rows(), rrows = right->get_output()->get_rows();
  const ColumnID key{6};   // l_long / r_long: few distinct values, several partners per row
  for (const auto condition : {PredicateCondition::LessThan, PredicateCondition::GreaterThanEquals, PredicateCondition::NotEquals}) {
    const auto holds = [&](int32_t x, int32_t y) {
      return condition == PredicateCondition::LessThan ? x < y : condition == PredicateCondition::GreaterThanEquals ? x >= y : x != y;
    };
    size_t inner = 0, semi = 0, anti = 0;
    for (const auto& l : lrows) {
      bool any = false;
      for (const auto& r : rrows) {
        if (variant_is_null(l[key]) || variant_is_null(r[key]) || std::get<int64_t>(l[key]) != std::get<int64_t>(r[key])) continue;
        if (!holds(std::get<int32_t>(l[0]), std::get<int32_t>(r[0]))) continue;
        ++inner;
        any = true;
      }
      semi += any;
      anti += !any;
    }
    const std::vector<OperatorJoinPredicate> secondary{{ColumnIDPair{ColumnID{0}, ColumnID{0}}, condition}};
    auto join = std::make_shared<JoinHash>(left, right, JoinMode::Inner, ColumnIDPair{key, key}, std::nullopt, secondary);
    join->execute();
    EXPECT_TRUE(join->get_output()->row_count() == inner);
    for (const auto& row : join->get_output()->get_rows()) {
      EXPECT_TRUE(cells_equal(row[key], row[lrows[0].size() + key]));
      EXPECT_TRUE(holds(std::get<int32_t>(row[0]), std::get<int32_t>(row[lrows[0].size()])));
    }
    auto semi_join = std::make_shared<JoinHash>(left, right, JoinMode::Semi, ColumnIDPair{key, key}, 2, secondary);
    semi_join->execute();
    EXPECT_TRUE(semi_join->get_output()->row_count() == semi);
    auto anti_join = std::make_shared<JoinHash>(left, right, JoinMode::AntiNullAsFalse, ColumnIDPair{key, key}, std::nullopt, secondary);
    anti_join->execute();
    EXPECT_TRUE(anti_join->get_output()->row_count() == anti);
    auto left_join = std::make_shared<JoinHash>(left, right, JoinMode::Left, ColumnIDPair{key, key}, std::nullopt, secondary);
    left_join->execute();
    EXPECT_TRUE(left_join->get_output()->row_count() == inner + anti);
    bool thrown = false;   // JoinHash::supports: no secondary predicates with AntiNullAsTrue
    try { auto j = std::make_shared<JoinHash>(left, right, JoinMode::AntiNullAsTrue, ColumnIDPair{key, key}, std::nullopt, secondary); j->execute(); }
    catch (const std::logic_error&) { thrown = true; }
    EXPECT_TRUE(thrown);
  }
}

static void test_aggregates_against_fixtures() {   // aggregate_test.cpp: test_output<>(input, aggregates, group by, expected)
  struct Case { std::string input; std::vector<AggregateDefinition> aggregates; std::vector<ColumnID> groupby; std::string expected; };
  const std::string d1 = "aggregateoperator/groupby_int_1gb_1agg/", d2 = "aggregateoperator/groupby_int_1gb_2agg/", d21 = "aggregateoperator/groupby_int_2gb_1agg/";
  const std::vector<Case> cases = {
      {d1 + "input.tbl", {{1, WindowFunction::Max}}, {0}, d1 + "max.tbl"},      {d1 + "input.tbl", {{1, WindowFunction::Min}}, {0}, d1 + "min.tbl"},
      {d1 + "input.tbl", {{1, WindowFunction::Sum}}, {0}, d1 + "sum.tbl"},      {d1 + "input.tbl", {{1, WindowFunction::Avg}}, {0}, d1 + "avg.tbl"},
      {d1 + "input.tbl", {{1, WindowFunction::Count}}, {0}, d1 + "count.tbl"},  {d1 + "input_null.tbl", {{1, WindowFunction::Sum}}, {0}, d1 + "sum_null.tbl"},
      {d1 + "input_null.tbl", {{1, WindowFunction::Avg}}, {0}, d1 + "avg_null.tbl"}, {d1 + "input_null.tbl", {{INVALID_COLUMN_ID, WindowFunction::Count}}, {0}, d1 + "count_star_null.tbl"},
      {d2 + "input.tbl", {{1, WindowFunction::Max}, {2, WindowFunction::Avg}}, {0}, d2 + "max_avg.tbl"},
      {d2 + "input.tbl", {{1, WindowFunction::Sum}, {2, WindowFunction::Sum}}, {0}, d2 + "sum_sum.tbl"},
      {d21 + "input.tbl", {{2, WindowFunction::Max}}, {0, 1}, d21 + "max.tbl"}, {d21 + "input.tbl", {{2, WindowFunction::Sum}}, {0, 1}, d21 + "sum.tbl"},
      {"aggregateoperator/groupby_string_1gb_1agg/input.tbl", {{1, WindowFunction::Sum}}, {0}, "aggregateoperator/groupby_string_1gb_1agg/sum.tbl"},
      {"aggregateoperator/groupby_string_1gb_1agg/input.tbl", {{1, WindowFunction::Avg}}, {0}, "aggregateoperator/groupby_string_1gb_1agg/avg.tbl"},
  };
  for (const auto encoding : {EncodingType::Unencoded, EncodingType::Dictionary}) {
    for (const auto& c : cases) {
      if (encoding == EncodingType::Unencoded && c.input.find("string") != std::string::npos) continue;   // unencoded strings: CPU path
      const auto wrapper = load_and_encode(c.input, 2, encoding);
      auto aggregate = std::make_shared<AggregateHash>(wrapper, c.aggregates, c.groupby);
      aggregate->execute();
      const bool ok = tables_equal_unordered(aggregate->get_output(), load_table(g_tbl + "/" + c.expected, 100));
      if (!ok) std::printf("  case %s -> %s\n", c.input.c_str(), c.expected.c_str());
      EXPECT_TRUE(ok);
    }
  }
  // CannotSumStringColumns (aggregate_test.cpp:232-241)
  const auto strings = load_and_encode("aggregateoperator/groupby_string_1gb_1agg/input.tbl", 2, EncodingType::Dictionary);
  bool thrown = false;
  try { auto a = std::make_shared<AggregateHash>(strings, std::vector<AggregateDefinition>{{0, WindowFunction::Sum}}, std::vector<ColumnID>{0}); a->execute(); }
  catch (const std::logic_error&) { thrown = true; }
  EXPECT_TRUE(thrown);
}

static void test_validate_visibility() {   // validate_visibility_test.cpp:45-131 (our_tid 2, snapshot 2) + a scan on its output
  struct Case { const char* name; TransactionID tid; CommitID begin, end; uint64_t rows; };
  const Case cases[] = {{"Impossible", 2, 2, 2, 0}, {"PastDelete", 42, 2, 2, 0}, {"Impossible2", 2, 4, 1, 0}, {"OwnDeleteUncommitted", 2, 1, 6, 0},
                        {"Impossible3", 50, 3, 1, 0}, {"OwnInsert", 2, 3, 3, 1}, {"PastInsertOrFutureDelete", 99, 2, 3, 1},
                        {"UncommittedInsertOrFutureInsert", 99, 3, 3, 0}};
  const auto make_table = [](size_t rows) {
    auto table = std::make_shared<Table>(TableColumnDefinitions{{"a", DataType::Int, false}, {"b", DataType::Int, false}}, TableType::Data, ChunkOffset{10});
    std::vector<int32_t> a(rows), b(rows);
    for (size_t i = 0; i < rows; ++i) { a[i] = 123 + static_cast<int32_t>(i); b[i] = 456; }
    table->append_chunk({std::make_shared<ValueSegment<int32_t>>(a, std::nullopt), std::make_shared<ValueSegment<int32_t>>(b, std::nullopt)}, std::make_shared<MvccData>(rows));
    return table;
  };
  for (const auto& c : cases) {
    const auto table = make_table(1);
    const auto& mvcc = table->get_chunk(0)->mvcc_data();
    mvcc->set_tid(0, c.tid); mvcc->set_begin_cid(0, c.begin); mvcc->set_end_cid(0, c.end);
    auto validate = std::make_shared<Validate>(wrap(table));
    validate->set_transaction_context(std::make_shared<TransactionContext>(2, 2));
    validate->execute();
    EXPECT_TRUE(validate->get_output()->row_count() == c.rows);
  }
  // all eight rows in one chunk: rows 5 and 6 survive; a TableScan runs on Validate's reference output
  const auto table = make_table(8);
  for (ChunkOffset i = 0; i < 8; ++i) {
    const auto& mvcc = table->get_chunk(0)->mvcc_data();
    mvcc->set_tid(i, cases[i].tid); mvcc->set_begin_cid(i, cases[i].begin); mvcc->set_end_cid(i, cases[i].end);
  }
  auto validate = std::make_shared<Validate>(wrap(table));
  validate->set_transaction_context(std::make_shared<TransactionContext>(2, 2));
  validate->execute();
  EXPECT_TRUE(column_ints(validate->get_output(), ColumnID{0}) == (std::vector<int32_t>{128, 129}));
  auto scan = std::make_shared<TableScan>(validate, ColumnID{0}, PredicateCondition::GreaterThan, AllTypeVariant{int32_t{128}});
  scan->execute();
  EXPECT_TRUE(column_ints(scan->get_output(), ColumnID{0}) == (std::vector<int32_t>{129}));
  // Validate on a reference table (the scan's output) and the entirely-visible shortcut on an immutable, committed chunk
  auto again = std::make_shared<Validate>(scan);
  again->set_transaction_context(std::make_shared<TransactionContext>(2, 2));
  again->execute();
  EXPECT_TRUE(column_ints(again->get_output(), ColumnID{0}) == (std::vector<int32_t>{129}));
  const auto old_table = make_table(5);
  for (ChunkOffset i = 0; i < 5; ++i) old_table->get_chunk(0)->mvcc_data()->set_begin_cid(i, 1);
  old_table->get_chunk(0)->mvcc_data()->max_begin_cid = 1;
  old_table->get_chunk(0)->set_immutable();
  auto shortcut = std::make_shared<Validate>(wrap(old_table));
  shortcut->set_transaction_context(std::make_shared<TransactionContext>(7, 3));
  shortcut->execute();
  EXPECT_TRUE(shortcut->get_output()->row_count() == 5);
  const auto out_segment = std::static_pointer_cast<ReferenceSegment>(shortcut->get_output()->get_chunk(0)->get_segment(0));
  EXPECT_TRUE(dynamic_cast<const EntireChunkPosList*>(out_segment->pos_list().get()) != nullptr);   // validate.cpp:282-284
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: host_tests <tbl directory>\n"); return 2; }
  g_tbl = argv[1];
  check_status(hy_init(0));
  run("TableScan.ScanOnCompressedSegments (+ out-of-range literals)", test_scan_on_compressed_segments);
  run("TableScan.ScanOnReferencedCompressedSegments", test_scan_on_referenced_segments);
  run("TableScan.SingleScan / DoubleScan / Between", test_single_and_double_scan);
  run("TableScan.ScanForNullValues", test_scan_for_null_values);
  run("TableScan.DictionarySegment<string>", test_string_dictionary_scan);
  run("TableScan.LikeOnDictionarySegments", test_like_on_dictionary_segments);
  run("TableScan.TypeMismatchThrowsLogicError", test_type_mismatch_throws);
  run("TableScan: literals of another type go through the lossless predicate cast", test_literals_of_other_types_are_cast_without_loss);
  run("JoinHash: small output PosLists are merged (1000 / 4000 rule)", test_join_output_chunks_are_merged);
  run("Validate.Visibility truth table, reference input, chunk shortcut", test_validate_visibility);
  run("JoinHash vs nested loop (Inner/Semi/AntiNullAsFalse/Left)", test_join_against_nested_loop);
  run("JoinHash on every pair of numeric key types vs nested loop", test_join_on_mixed_numeric_key_types);
  run("JoinHash on string keys (join ids) vs nested loop", test_join_on_string_keys);
  run("JoinHash with secondary predicates vs nested loop", test_join_with_secondary_predicates);
  run("AggregateHash vs .tbl fixtures (+ CannotSumStringColumns)", test_aggregates_against_fixtures);
  run("ScanProjectAggregate (fused pass) with and without Validate", test_scan_project_aggregate);
  run("StarJoinAggregate (hy_star_join_aggregate) against nested loops", test_star_join_aggregate);
  hy_shutdown();
  std::printf("%s\n", g_failures ? "HOST TESTS FAILED" : "HOST TESTS PASSED");
  return g_failures ? 1 : 0;
}
