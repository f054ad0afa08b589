// hyrise_host.hpp -- C++ mirror of the slice of Hyrise's operator/storage interface that the hot path touches, written
// against the C ABI (include/hyrise_amd.h).  Hyrise itself cannot be built in this environment;
// this header lets the three operators be driven -- and tested -- exactly the way the reference's own tests drive them:
//
//     auto table   = load_table("int_float.tbl", ChunkOffset{2});            src/lib/utils/load_table.cpp:57-96
//     ChunkEncoder::encode_all_chunks(table, EncodingType::Dictionary);      src/lib/storage/chunk_encoder.hpp
//     auto wrapper = std::make_shared<TableWrapper>(table); wrapper->execute();
//     auto scan    = std::make_shared<TableScan>(wrapper, ColumnID{0}, PredicateCondition::GreaterThanEquals, 1234);
//     scan->execute();  scan->get_output();                                  operators/abstract_operator.hpp:133,143
//
// Names, argument meaning and error behaviour follow the reference: operators derive from AbstractReadOnlyOperator and
// implement `_on_execute()` (abstract_read_only_operator.hpp:20-22); scans and joins return TableType::References
// tables whose segments share PosLists (table_scan.cpp:207-210); failures throw std::logic_error (utils/assert.hpp:48-70).
// There is NO CPU implementation behind these classes: what the device library reports as HY_ERR_UNSUPPORTED (the
// shapes for which the real adapter keeps Hyrise's stock operator) surfaces as std::logic_error here.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/hyrise_amd.h"

namespace hyrise_amd {

// ---- types.hpp / all_type_variant.hpp ---------------------------------------------------------------------------------
using ChunkID = uint32_t;
using ChunkOffset = uint32_t;
using ColumnID = uint16_t;
constexpr ColumnID INVALID_COLUMN_ID = 0xFFFF;

enum class DataType : uint8_t { Null, Int, Long, Float, Double, String };
enum class PredicateCondition : uint8_t {
  Equals, NotEquals, LessThan, LessThanEquals, GreaterThan, GreaterThanEquals, BetweenInclusive, BetweenLowerExclusive,
  BetweenUpperExclusive, BetweenExclusive, In, NotIn, Like, NotLike, LikeInsensitive, NotLikeInsensitive, IsNull, IsNotNull
};
enum class JoinMode : uint8_t { Inner, Left, Right, FullOuter, Cross, Semi, AntiNullAsTrue, AntiNullAsFalse };
enum class WindowFunction : uint8_t { Min, Max, Sum, Avg, Count, CountDistinct, StandardDeviationSample, Any };
enum class EncodingType : uint8_t { Unencoded, Dictionary, FrameOfReference, LZ4 };
enum class TableType : uint8_t { Data, References };
enum class SortMode : uint8_t { AscendingNullsFirst, DescendingNullsFirst, AscendingNullsLast, DescendingNullsLast };   // types.hpp:219
struct SortColumnDefinition {   // types.hpp:245-251
  explicit SortColumnDefinition(ColumnID init_column, SortMode init_sort_mode = SortMode::AscendingNullsFirst) : column(init_column), sort_mode(init_sort_mode) {}
  ColumnID column;
  SortMode sort_mode;
};

struct NullValue {};
using AllTypeVariant = std::variant<NullValue, int32_t, int64_t, float, double, std::string>;
inline bool variant_is_null(const AllTypeVariant& v) { return v.index() == 0; }
inline DataType data_type_from_all_type_variant(const AllTypeVariant& v) { return static_cast<DataType>(v.index()); }

struct RowID {
  ChunkID chunk_id;
  ChunkOffset chunk_offset;
  bool is_null() const { return chunk_offset == 0xFFFFFFFFu; }
  bool operator==(const RowID& o) const { return chunk_id == o.chunk_id && chunk_offset == o.chunk_offset; }
};
static_assert(sizeof(RowID) == sizeof(hy_row_id), "RowID must be the ABI's 8-byte {chunk_id, chunk_offset}");
constexpr RowID NULL_ROW_ID{0xFFFFFFFFu, 0xFFFFFFFFu};

[[noreturn]] inline void Fail(const std::string& message) { throw std::logic_error(message); }
inline void Assert(bool condition, const char* message) { if (!condition) Fail(message); }   // (no std::string built on the passing path: thousands of ReferenceSegments per output table assert)
inline void Assert(bool condition, const std::string& message) { if (!condition) Fail(message); }
inline void check_status(hy_status status) { if (status != HY_OK) Fail(std::string("hyrise_amd: ") + hy_last_error()); }

// ---- storage ---------------------------------------------------------------------------------------------------------
class AbstractPosList {
 public:
  virtual ~AbstractPosList() = default;
  virtual size_t size() const = 0;
  virtual RowID operator[](size_t index) const = 0;
  virtual bool references_single_chunk() const = 0;
  virtual ChunkID common_chunk_id() const = 0;
};

class RowIDPosList : public AbstractPosList {   // pos_lists/row_id_pos_list.hpp:22-143
 public:
  RowIDPosList() = default;
  explicit RowIDPosList(std::vector<RowID> init) : rows(std::move(init)) {}
  size_t size() const override { return rows.size(); }
  RowID operator[](size_t index) const override { return rows[index]; }
  void guarantee_single_chunk() { _single = true; }
  bool references_single_chunk() const override { return _single; }
  ChunkID common_chunk_id() const override { return rows.empty() ? 0xFFFFFFFFu : rows[0].chunk_id; }
  std::vector<RowID> rows;

 private:
  bool _single = false;
};

class EntireChunkPosList : public AbstractPosList {   // pos_lists/entire_chunk_pos_list.hpp:8-44
 public:
  EntireChunkPosList(ChunkID chunk_id, ChunkOffset size) : _chunk_id(chunk_id), _size(size) {}
  size_t size() const override { return _size; }
  RowID operator[](size_t index) const override { return RowID{_chunk_id, static_cast<ChunkOffset>(index)}; }
  bool references_single_chunk() const override { return true; }
  ChunkID common_chunk_id() const override { return _chunk_id; }

 private:
  ChunkID _chunk_id;
  ChunkOffset _size;
};

// ---- PosLists that stay in HBM ------------------------------------------------------------------------------------------------
// The reference's PosLists are polymorphic (AbstractPosList, pos_lists/abstract_pos_list.hpp:18-75; ReferenceSegment holds a
// shared_ptr<const AbstractPosList>, reference_segment.hpp:36-38): an operator that understands a subclass reads it directly, everybody
// else goes through size() / operator[] / begin().  DevicePosList is that subclass for this library: the RowIDs lie in a block of the
// library's result-buffer pool (hy_result_pool_*), the operators below hand the device pointer to the next call (a scan's PosLists as the
// reference column of a join, a join's as the input of an aggregate) and nothing crosses the host link; code that indexes the list gets
// a host copy made on first use.  The block goes back to the pool when the last PosList in it dies.
struct DeviceBlock {
  explicit DeviceBlock(void* init) : ptr(init) {}
  DeviceBlock(const DeviceBlock&) = delete;
  DeviceBlock& operator=(const DeviceBlock&) = delete;
  ~DeviceBlock() { if (ptr) (void)hy_result_pool_release(ptr); }
  static std::shared_ptr<DeviceBlock> acquire(uint64_t bytes) {
    void* p = nullptr;
    check_status(hy_result_pool_acquire(bytes, &p));
    return std::make_shared<DeviceBlock>(p);
  }
  // ONE transfer for every PosList of the block (a scan's 916 lists, a join's 458): the lists then index this copy
  void prefetch_to_host(size_t rows) const {
    std::call_once(_once, [&] {
      _host.resize(rows);
      if (rows) check_status(hy_memcpy_d2h(_host.data(), ptr, rows * sizeof(RowID)));
      _published.store(rows ? _host.data() : nullptr, std::memory_order_release);
    });
  }
  const RowID* host_copy() const { return _published.load(std::memory_order_acquire); }
  void* ptr;

 private:
  mutable std::once_flag _once;
  mutable std::vector<RowID> _host;
  mutable std::atomic<const RowID*> _published{nullptr};
};

// Hands every operator's device results to the next operator without a host copy (default).  false: the operators ask the library for
// HY_MEM_HOST results, as rounds 1-5 did -- tests compare the two.
inline bool& device_resident_results() { static bool enabled = true; return enabled; }

class DevicePosList : public AbstractPosList {
 public:
  static constexpr ChunkID NO_COMMON_CHUNK = 0xFFFFFFFFu;
  DevicePosList(std::shared_ptr<const DeviceBlock> block, const hy_row_id* rows, size_t size, bool single_chunk = false, ChunkID common_chunk = NO_COMMON_CHUNK)
      : _block(std::move(block)), _rows(rows), _size(size), _single(single_chunk), _common(common_chunk) {}
  size_t size() const override { return _size; }
  // (a handful of lookups -- an aggregate's representative rows -- fetch eight bytes each; a reader that keeps indexing gets the copy)
  RowID operator[](size_t index) const override {
    if (!_copied.load(std::memory_order_acquire) && !_block->host_copy() && _single_fetches.fetch_add(1, std::memory_order_relaxed) < 16) {
      RowID row{};
      check_status(hy_memcpy_d2h(&row, _rows + index, sizeof(RowID)));
      return row;
    }
    return host_rows()[index];
  }
  bool references_single_chunk() const override { return _single; }
  ChunkID common_chunk_id() const override { return _single ? _common : NO_COMMON_CHUNK; }
  const hy_row_id* device_data() const { return _rows; }
  const std::shared_ptr<const DeviceBlock>& block() const { return _block; }
  // begin() of the reference's interface: the rows in host memory, copied on first use (from the block's copy if somebody prefetched it)
  const RowID* host_rows() const {
    if (const auto* all = _block->host_copy()) return all + (_rows - static_cast<const hy_row_id*>(_block->ptr));
    std::call_once(_once, [&] { _host.resize(_size); if (_size) check_status(hy_memcpy_d2h(_host.data(), _rows, _size * sizeof(RowID))); _copied.store(true, std::memory_order_release); });
    return _host.data();
  }

 private:
  std::shared_ptr<const DeviceBlock> _block;
  const hy_row_id* _rows;
  size_t _size;
  bool _single;
  ChunkID _common;
  mutable std::once_flag _once;
  mutable std::vector<RowID> _host;
  mutable std::atomic<bool> _copied{false};
  mutable std::atomic<uint32_t> _single_fetches{0};
};

class Table;

class AbstractSegment {
 public:
  explicit AbstractSegment(DataType type) : _data_type(type) {}
  virtual ~AbstractSegment() = default;
  virtual ChunkOffset size() const = 0;
  virtual AllTypeVariant operator[](ChunkOffset offset) const = 0;   // slow, test-side access like the reference's
  DataType data_type() const { return _data_type; }

 private:
  DataType _data_type;
};

template <typename T> constexpr DataType data_type_of() {
  if constexpr (std::is_same_v<T, int32_t>) return DataType::Int;
  else if constexpr (std::is_same_v<T, int64_t>) return DataType::Long;
  else if constexpr (std::is_same_v<T, float>) return DataType::Float;
  else if constexpr (std::is_same_v<T, double>) return DataType::Double;
  else return DataType::String;
}

inline std::vector<uint64_t> pack_null_words(const std::vector<bool>& nulls) {   // libstdc++ vector<bool> word order
  std::vector<uint64_t> words((nulls.size() + 63) / 64, 0);
  for (size_t i = 0; i < nulls.size(); ++i) if (nulls[i]) words[i / 64] |= uint64_t{1} << (i % 64);
  return words;
}

template <typename T>
class ValueSegment : public AbstractSegment {   // storage/value_segment.hpp:15-90
 public:
  ValueSegment(std::vector<T> values, std::optional<std::vector<bool>> nulls) : AbstractSegment(data_type_of<T>()), _values(std::move(values)), _nulls(std::move(nulls)) {
    if (_nulls) _null_words = pack_null_words(*_nulls);
  }
  ChunkOffset size() const override { return static_cast<ChunkOffset>(_values.size()); }
  bool is_nullable() const { return _nulls.has_value(); }
  const std::vector<T>& values() const { return _values; }
  const std::vector<uint64_t>& null_words() const { return _null_words; }
  bool is_null(ChunkOffset offset) const { return _nulls && (*_nulls)[offset]; }
  AllTypeVariant operator[](ChunkOffset offset) const override { if (is_null(offset)) return NullValue{}; return _values[offset]; }

 private:
  std::vector<T> _values;
  std::optional<std::vector<bool>> _nulls;
  std::vector<uint64_t> _null_words;
};

// FixedWidthIntegerVector<u8/u16/u32> (fixed_width_integer_compressor.cpp:33-44)
struct CompressedVector {
  uint32_t width = 4;
  std::vector<uint8_t> bytes;
  static CompressedVector compress(const std::vector<uint32_t>& values, uint32_t max_value) {
    CompressedVector out;
    out.width = max_value <= 0xFF ? 1 : max_value <= 0xFFFF ? 2 : 4;
    out.bytes.resize(values.size() * out.width + 16);
    for (size_t i = 0; i < values.size(); ++i) std::memcpy(out.bytes.data() + i * out.width, &values[i], out.width);
    return out;
  }
  uint32_t get(size_t i) const { uint32_t v = 0; std::memcpy(&v, bytes.data() + i * width, width); return v; }
};

template <typename T>
class DictionarySegment : public AbstractSegment {   // storage/dictionary_segment.hpp:19-91
 public:
  DictionarySegment(std::vector<T> dictionary, CompressedVector attribute_vector, ChunkOffset size)
      : AbstractSegment(data_type_of<T>()), _dictionary(std::move(dictionary)), _attribute_vector(std::move(attribute_vector)), _size(size) {}
  ChunkOffset size() const override { return _size; }
  const std::vector<T>& dictionary() const { return _dictionary; }
  const CompressedVector& attribute_vector() const { return _attribute_vector; }
  uint32_t unique_values_count() const { return static_cast<uint32_t>(_dictionary.size()); }
  uint32_t lower_bound(const T& value) const {   // dictionary_segment.cpp:94-106
    const auto it = std::lower_bound(_dictionary.begin(), _dictionary.end(), value);
    return it == _dictionary.end() ? HY_INVALID_VALUE_ID : static_cast<uint32_t>(it - _dictionary.begin());
  }
  uint32_t upper_bound(const T& value) const {   // :108-119
    const auto it = std::upper_bound(_dictionary.begin(), _dictionary.end(), value);
    return it == _dictionary.end() ? HY_INVALID_VALUE_ID : static_cast<uint32_t>(it - _dictionary.begin());
  }
  AllTypeVariant operator[](ChunkOffset offset) const override {
    const auto vid = _attribute_vector.get(offset);
    if (vid >= _dictionary.size()) return NullValue{};
    return _dictionary[vid];
  }

 private:
  std::vector<T> _dictionary;
  CompressedVector _attribute_vector;
  ChunkOffset _size;
};

class FrameOfReferenceSegment : public AbstractSegment {   // storage/frame_of_reference_segment.hpp:37-98
 public:
  FrameOfReferenceSegment(std::vector<int32_t> minima, CompressedVector offsets, std::optional<std::vector<bool>> nulls, ChunkOffset size)
      : AbstractSegment(DataType::Int), _minima(std::move(minima)), _offsets(std::move(offsets)), _nulls(std::move(nulls)), _size(size) {
    if (_nulls) _null_words = pack_null_words(*_nulls);
  }
  ChunkOffset size() const override { return _size; }
  const std::vector<int32_t>& block_minima() const { return _minima; }
  const CompressedVector& offset_values() const { return _offsets; }
  bool has_nulls() const { return _nulls.has_value(); }
  const std::vector<uint64_t>& null_words() const { return _null_words; }
  AllTypeVariant operator[](ChunkOffset offset) const override {
    if (_nulls && (*_nulls)[offset]) return NullValue{};
    return static_cast<int32_t>(_offsets.get(offset) + static_cast<uint32_t>(_minima[offset / HY_FOR_BLOCK_SIZE]));
  }

 private:
  std::vector<int32_t> _minima;
  CompressedVector _offsets;
  std::optional<std::vector<bool>> _nulls;
  std::vector<uint64_t> _null_words;
  ChunkOffset _size;
};

// LZ4Segment<T> of a numeric type (storage/lz4_segment.hpp:25-118): the values' bytes in blocks of BLOCK_SIZE, every block compressed on its
// own.  The device path takes the blocks as they are (HY_ENC_LZ4 + hy_lz4_blocks) and decompresses them with a kernel.
template <typename T>
class LZ4Segment : public AbstractSegment {
 public:
  static constexpr size_t BLOCK_SIZE = 16384;   // lz4_encoder.hpp:61
  LZ4Segment(std::vector<std::vector<char>> blocks, std::optional<std::vector<bool>> nulls, std::vector<char> dictionary, size_t block_size, size_t last_block_size, ChunkOffset size)
      : AbstractSegment(data_type_of<T>()), _blocks(std::move(blocks)), _nulls(std::move(nulls)), _dictionary(std::move(dictionary)), _block_size(block_size),
        _last_block_size(last_block_size), _size(size) {
    if (_nulls) _null_words = pack_null_words(*_nulls);
    for (const auto& block : _blocks) { _block_pointers.push_back(block.data()); _block_bytes.push_back(static_cast<uint32_t>(block.size())); }
    _descriptor.blocks = _block_pointers.data();
    _descriptor.block_bytes = _block_bytes.data();
    _descriptor.block_count = static_cast<uint32_t>(_blocks.size());
    _descriptor.block_size = static_cast<uint32_t>(_block_size);
    _descriptor.last_block_size = static_cast<uint32_t>(_last_block_size);
    _descriptor.dictionary_bytes = static_cast<uint32_t>(_dictionary.size());
    _descriptor.dictionary = _dictionary.empty() ? nullptr : _dictionary.data();
  }
  ChunkOffset size() const override { return _size; }
  AllTypeVariant operator[](ChunkOffset offset) const override {   // (the mirror's own decoder: decompress(), lz4_segment.cpp:125-136)
    if (_nulls && (*_nulls)[offset]) return NullValue{};
    return decompress()[offset];
  }
  std::vector<T> decompress() const {
    std::vector<char> bytes;
    for (size_t b = 0; b < _blocks.size(); ++b) {
      const auto& in = _blocks[b];
      const size_t start = bytes.size();
      size_t i = 0;
      while (i < in.size()) {
        const auto token = static_cast<uint8_t>(in[i++]);
        size_t literals = token >> 4;
        if (literals == 15) { uint8_t extra; do { extra = static_cast<uint8_t>(in[i++]); literals += extra; } while (extra == 255); }
        bytes.insert(bytes.end(), in.begin() + i, in.begin() + i + literals);
        i += literals;
        if (i >= in.size()) break;
        const size_t offset = static_cast<uint8_t>(in[i]) | static_cast<size_t>(static_cast<uint8_t>(in[i + 1])) << 8;
        i += 2;
        size_t length = (token & 15) + 4;
        if ((token & 15) == 15) { uint8_t extra; do { extra = static_cast<uint8_t>(in[i++]); length += extra; } while (extra == 255); }
        for (size_t k = 0; k < length; ++k) {
          const ptrdiff_t from = static_cast<ptrdiff_t>(bytes.size() - start) - static_cast<ptrdiff_t>(offset);
          bytes.push_back(from >= 0 ? bytes[start + from] : _dictionary[_dictionary.size() + from]);
        }
      }
    }
    std::vector<T> values(_size);
    std::memcpy(values.data(), bytes.data(), std::min(bytes.size(), sizeof(T) * _size));
    return values;
  }
  const hy_lz4_blocks& descriptor() const { return _descriptor; }
  bool has_nulls() const { return _nulls.has_value(); }
  const std::vector<uint64_t>& null_words() const { return _null_words; }

 private:
  std::vector<std::vector<char>> _blocks;
  std::optional<std::vector<bool>> _nulls;
  std::vector<char> _dictionary;
  size_t _block_size, _last_block_size;
  ChunkOffset _size;
  std::vector<uint64_t> _null_words;
  std::vector<const void*> _block_pointers;
  std::vector<uint32_t> _block_bytes;
  hy_lz4_blocks _descriptor{};
};

class ReferenceSegment : public AbstractSegment {   // storage/reference_segment.hpp:20-48
 public:
  ReferenceSegment(std::shared_ptr<const Table> table, ColumnID column_id, std::shared_ptr<const AbstractPosList> pos_list);
  ChunkOffset size() const override { return static_cast<ChunkOffset>(_pos_list->size()); }
  const std::shared_ptr<const Table>& referenced_table() const { return _table; }
  ColumnID referenced_column_id() const { return _column_id; }
  const std::shared_ptr<const AbstractPosList>& pos_list() const { return _pos_list; }
  AllTypeVariant operator[](ChunkOffset offset) const override;

 private:
  std::shared_ptr<const Table> _table;
  ColumnID _column_id;
  std::shared_ptr<const AbstractPosList> _pos_list;
};

using Segments = std::vector<std::shared_ptr<AbstractSegment>>;

using TransactionID = uint32_t;
using CommitID = uint32_t;
constexpr CommitID MAX_COMMIT_ID = 0xFFFFFFFFu - 1;        // storage/mvcc_data.hpp
constexpr TransactionID INVALID_TRANSACTION_ID = 0;

struct MvccData {   // storage/mvcc_data.hpp: three arrays, one entry per row of the chunk
  explicit MvccData(size_t size, CommitID begin_commit_id = MAX_COMMIT_ID)
      : tids(size, INVALID_TRANSACTION_ID), begin_cids(size, begin_commit_id), end_cids(size, MAX_COMMIT_ID), max_begin_cid(begin_commit_id) {}
  TransactionID get_tid(ChunkOffset row) const { return tids[row]; }
  CommitID get_begin_cid(ChunkOffset row) const { return begin_cids[row]; }
  CommitID get_end_cid(ChunkOffset row) const { return end_cids[row]; }
  void set_tid(ChunkOffset row, TransactionID tid) { tids[row] = tid; }
  void set_begin_cid(ChunkOffset row, CommitID cid) { begin_cids[row] = cid; }
  void set_end_cid(ChunkOffset row, CommitID cid) { end_cids[row] = cid; }
  std::vector<TransactionID> tids;
  std::vector<CommitID> begin_cids, end_cids;
  CommitID max_begin_cid;
};

class Chunk {   // storage/chunk.hpp:38-218
 public:
  static constexpr ChunkOffset DEFAULT_SIZE = 65535;
  explicit Chunk(Segments segments) : _segments(std::move(segments)) {}
  Chunk(Segments segments, std::shared_ptr<MvccData> mvcc_data) : _segments(std::move(segments)), _mvcc_data(std::move(mvcc_data)) {}
  bool has_mvcc_data() const { return _mvcc_data != nullptr; }
  const std::shared_ptr<MvccData>& mvcc_data() const { return _mvcc_data; }
  bool is_mutable() const { return _is_mutable; }
  void set_immutable() { _is_mutable = false; }
  uint32_t invalid_row_count() const { return _invalid_row_count; }
  void increase_invalid_row_count(uint32_t count) { _invalid_row_count += count; }
  ChunkOffset size() const { return _segments.empty() ? 0 : _segments[0]->size(); }
  const std::shared_ptr<AbstractSegment>& get_segment(ColumnID column_id) const { return _segments.at(column_id); }
  void replace_segment(ColumnID column_id, std::shared_ptr<AbstractSegment> segment) { _segments.at(column_id) = std::move(segment); }
  ColumnID column_count() const { return static_cast<ColumnID>(_segments.size()); }
  // chunk.hpp:160-176: the scan of a column the chunk is flagged as sorted by takes the SortedSegmentSearch path (HY_SORT_* on the descriptor)
  const std::vector<SortColumnDefinition>& individually_sorted_by() const { return _sorted_by; }
  void set_individually_sorted_by(const SortColumnDefinition& sorted_by) { _sorted_by = {sorted_by}; }
  void set_individually_sorted_by(const std::vector<SortColumnDefinition>& sorted_by) { _sorted_by = sorted_by; }

 private:
  Segments _segments;
  std::vector<SortColumnDefinition> _sorted_by;
  std::shared_ptr<MvccData> _mvcc_data;
  bool _is_mutable = true;
  uint32_t _invalid_row_count = 0;
};

struct TableColumnDefinition {
  std::string name;
  DataType data_type;
  bool nullable;
};
using TableColumnDefinitions = std::vector<TableColumnDefinition>;

struct ColumnCache;   // device-resident columns of one table (residency cache)

class Table : public std::enable_shared_from_this<Table> {   // storage/table.hpp
 public:
  Table(TableColumnDefinitions definitions, TableType type, ChunkOffset target_chunk_size = Chunk::DEFAULT_SIZE)
      : _definitions(std::move(definitions)), _type(type), _target_chunk_size(target_chunk_size) {}
  Table(TableColumnDefinitions definitions, TableType type, std::vector<std::shared_ptr<Chunk>> chunks)
      : _definitions(std::move(definitions)), _type(type), _target_chunk_size(Chunk::DEFAULT_SIZE), _chunks(std::move(chunks)) {}
  ~Table();
  const TableColumnDefinitions& column_definitions() const { return _definitions; }
  ColumnID column_count() const { return static_cast<ColumnID>(_definitions.size()); }
  DataType column_data_type(ColumnID id) const { return _definitions.at(id).data_type; }
  bool column_is_nullable(ColumnID id) const { return _definitions.at(id).nullable; }
  const std::string& column_name(ColumnID id) const { return _definitions.at(id).name; }
  TableType type() const { return _type; }
  ChunkID chunk_count() const { return static_cast<ChunkID>(_chunks.size()); }
  ChunkOffset target_chunk_size() const { return _target_chunk_size; }
  const std::shared_ptr<Chunk>& get_chunk(ChunkID id) const { return _chunks.at(id); }
  uint64_t row_count() const { uint64_t n = 0; for (const auto& c : _chunks) n += c->size(); return n; }
  void append_chunk(Segments segments) { _chunks.push_back(std::make_shared<Chunk>(std::move(segments))); }
  void append_chunk(Segments segments, std::shared_ptr<MvccData> mvcc_data) { _chunks.push_back(std::make_shared<Chunk>(std::move(segments), std::move(mvcc_data))); }
  // Table::append(row): rows are collected and cut into ValueSegments of target_chunk_size rows by finalize().
  void append(std::vector<AllTypeVariant> row) { _pending.push_back(std::move(row)); if (_pending.size() == _target_chunk_size) finalize(); }
  void finalize();
  AllTypeVariant get_value(ColumnID column_id, uint64_t row) const {
    for (const auto& chunk : _chunks) { if (row < chunk->size()) return (*chunk->get_segment(column_id))[static_cast<ChunkOffset>(row)]; row -= chunk->size(); }
    Fail("row index out of range");
  }
  std::vector<std::vector<AllTypeVariant>> get_rows() const {
    std::vector<std::vector<AllTypeVariant>> rows;
    for (const auto& chunk : _chunks)
      for (ChunkOffset r = 0; r < chunk->size(); ++r) {
        std::vector<AllTypeVariant> row;
        for (ColumnID c = 0; c < column_count(); ++c) row.push_back((*chunk->get_segment(c))[r]);
        rows.push_back(std::move(row));
      }
    return rows;
  }
  mutable std::shared_ptr<ColumnCache> device_columns;

 private:
  TableColumnDefinitions _definitions;
  TableType _type;
  ChunkOffset _target_chunk_size;
  std::vector<std::shared_ptr<Chunk>> _chunks;
  std::vector<std::vector<AllTypeVariant>> _pending;
};

inline ReferenceSegment::ReferenceSegment(std::shared_ptr<const Table> table, ColumnID column_id, std::shared_ptr<const AbstractPosList> pos_list)
    : AbstractSegment(table->column_data_type(column_id)), _table(std::move(table)), _column_id(column_id), _pos_list(std::move(pos_list)) {
  Assert(_table->type() == TableType::Data, "ReferenceSegments must not reference reference tables (table_scan.cpp:140-148)");
}
inline AllTypeVariant ReferenceSegment::operator[](ChunkOffset offset) const {
  const RowID row = (*_pos_list)[offset];
  if (row.is_null()) return NullValue{};
  return (*_table->get_chunk(row.chunk_id)->get_segment(_column_id))[row.chunk_offset];
}

template <typename T>
std::shared_ptr<AbstractSegment> make_value_segment(const std::vector<std::vector<AllTypeVariant>>& rows, ColumnID column, bool nullable) {
  std::vector<T> values;
  std::vector<bool> nulls;
  for (const auto& row : rows) {
    const bool is_null = variant_is_null(row[column]);
    Assert(!is_null || nullable, "NULL in a non-nullable column");
    nulls.push_back(is_null);
    values.push_back(is_null ? T{} : std::get<T>(row[column]));
  }
  return std::make_shared<ValueSegment<T>>(std::move(values), nullable ? std::optional<std::vector<bool>>(std::move(nulls)) : std::nullopt);
}

inline void Table::finalize() {
  if (_pending.empty()) return;
  Segments segments;
  for (ColumnID c = 0; c < column_count(); ++c) {
    switch (_definitions[c].data_type) {
      case DataType::Int: segments.push_back(make_value_segment<int32_t>(_pending, c, _definitions[c].nullable)); break;
      case DataType::Long: segments.push_back(make_value_segment<int64_t>(_pending, c, _definitions[c].nullable)); break;
      case DataType::Float: segments.push_back(make_value_segment<float>(_pending, c, _definitions[c].nullable)); break;
      case DataType::Double: segments.push_back(make_value_segment<double>(_pending, c, _definitions[c].nullable)); break;
      default: segments.push_back(make_value_segment<std::string>(_pending, c, _definitions[c].nullable)); break;
    }
  }
  _chunks.push_back(std::make_shared<Chunk>(std::move(segments)));
  _pending.clear();
}

// ---- load_table (utils/load_table.cpp:22-96) ----------------------------------------------------------------------------
inline std::vector<std::string> split_string_by_delimiter(const std::string& line, char delimiter) {
  std::vector<std::string> out;
  std::stringstream stream(line);
  std::string cell;
  while (std::getline(stream, cell, delimiter)) out.push_back(cell);
  if (!line.empty() && line.back() == delimiter) out.emplace_back();
  return out;
}

inline std::shared_ptr<Table> load_table(const std::string& file_name, ChunkOffset chunk_size = Chunk::DEFAULT_SIZE) {
  std::ifstream infile(file_name);
  Assert(infile.is_open(), "load_table: Could not find file '" + file_name + "'.");
  std::string line;
  std::getline(infile, line);
  const auto names = split_string_by_delimiter(line, '|');
  std::getline(infile, line);
  const auto types = split_string_by_delimiter(line, '|');
  TableColumnDefinitions definitions;
  for (size_t i = 0; i < names.size(); ++i) {
    const auto parts = split_string_by_delimiter(types[i], '_');
    const bool nullable = parts.size() > 1 && parts[1] == "null";
    DataType type;
    if (parts[0] == "int") type = DataType::Int;
    else if (parts[0] == "long") type = DataType::Long;
    else if (parts[0] == "float") type = DataType::Float;
    else if (parts[0] == "double") type = DataType::Double;
    else if (parts[0] == "string") type = DataType::String;
    else Fail("Invalid data type '" + parts[0] + "' for column '" + names[i] + "'.");
    definitions.push_back({names[i], type, nullable});
  }
  auto table = std::make_shared<Table>(definitions, TableType::Data, chunk_size);
  while (std::getline(infile, line)) {
    auto cells = split_string_by_delimiter(line, '|');
    cells.resize(definitions.size());
    std::vector<AllTypeVariant> row;
    for (size_t c = 0; c < definitions.size(); ++c) {
      if (definitions[c].nullable && cells[c] == "null") { row.emplace_back(NullValue{}); continue; }
      switch (definitions[c].data_type) {
        case DataType::Int: row.emplace_back(static_cast<int32_t>(std::stol(cells[c]))); break;
        case DataType::Long: row.emplace_back(static_cast<int64_t>(std::stoll(cells[c]))); break;
        case DataType::Float: row.emplace_back(std::stof(cells[c])); break;
        case DataType::Double: row.emplace_back(std::stod(cells[c])); break;
        default: row.emplace_back(cells[c]); break;
      }
    }
    table->append(std::move(row));
  }
  table->finalize();
  return table;
}

// ---- ChunkEncoder (dictionary_encoder.hpp:33-103, frame_of_reference_encoder.hpp:25-122) --------------------------------
struct ChunkEncoder {
  template <typename T>
  static std::shared_ptr<AbstractSegment> encode_dictionary(const ValueSegment<T>& segment) {
    std::vector<T> dictionary;
    for (ChunkOffset i = 0; i < segment.size(); ++i) if (!segment.is_null(i)) dictionary.push_back(segment.values()[i]);
    std::sort(dictionary.begin(), dictionary.end());
    dictionary.erase(std::unique(dictionary.begin(), dictionary.end()), dictionary.end());
    const auto null_value_id = static_cast<uint32_t>(dictionary.size());
    std::vector<uint32_t> ids(segment.size());
    for (ChunkOffset i = 0; i < segment.size(); ++i) {
      ids[i] = segment.is_null(i) ? null_value_id
                                  : static_cast<uint32_t>(std::lower_bound(dictionary.begin(), dictionary.end(), segment.values()[i]) - dictionary.begin());
    }
    return std::make_shared<DictionarySegment<T>>(std::move(dictionary), CompressedVector::compress(ids, null_value_id), segment.size());
  }
  static std::shared_ptr<AbstractSegment> encode_frame_of_reference(const ValueSegment<int32_t>& segment) {
    const auto n = segment.size();
    std::vector<int32_t> minima;
    std::vector<uint32_t> offsets(n);
    std::vector<bool> nulls(n, false);
    bool any_null = false;
    uint32_t max_offset = 0;
    for (ChunkOffset begin = 0; begin < n; begin += HY_FOR_BLOCK_SIZE) {
      const auto end = std::min<ChunkOffset>(n, begin + HY_FOR_BLOCK_SIZE);
      int32_t minimum = INT32_MAX;
      for (auto i = begin; i < end; ++i) {
        nulls[i] = segment.is_null(i);
        any_null |= nulls[i];
        if (!nulls[i]) minimum = std::min(minimum, segment.values()[i]);
      }
      minima.push_back(minimum);
      for (auto i = begin; i < end; ++i) {
        const int32_t value = nulls[i] ? minimum : segment.values()[i];
        offsets[i] = static_cast<uint32_t>(value) - static_cast<uint32_t>(minimum);
        max_offset = std::max(max_offset, offsets[i]);
      }
    }
    return std::make_shared<FrameOfReferenceSegment>(std::move(minima), CompressedVector::compress(offsets, max_offset),
                                                     any_null ? std::optional<std::vector<bool>>(std::move(nulls)) : std::nullopt, n);
  }
  // One LZ4 block (lz4_Block_format.md) by a greedy matcher over 4-byte words -- enough to produce what liblz4 would also accept; the
  // mirror has no zstd dictionary trainer, so its blocks are compressed without one (each still decompresses on its own).
  static std::vector<char> lz4_compress_block(const char* in, size_t n) {
    std::vector<char> out;
    std::vector<int32_t> last(1 << 12, -1);
    size_t anchor = 0, i = 0;
    auto emit = [&](size_t literal_end, size_t offset, size_t length) {   // length 0: the closing literals
      const size_t literals = literal_end - anchor;
      out.push_back(static_cast<char>((std::min<size_t>(literals, 15) << 4) | (length ? std::min<size_t>(length - 4, 15) : 0)));
      if (literals >= 15) { size_t rest = literals - 15; while (rest >= 255) { out.push_back(static_cast<char>(255)); rest -= 255; } out.push_back(static_cast<char>(rest)); }
      out.insert(out.end(), in + anchor, in + literal_end);
      if (!length) return;
      out.push_back(static_cast<char>(offset & 0xFF));
      out.push_back(static_cast<char>(offset >> 8));
      if (length - 4 >= 15) { size_t rest = length - 4 - 15; while (rest >= 255) { out.push_back(static_cast<char>(255)); rest -= 255; } out.push_back(static_cast<char>(rest)); }
    };
    while (i + 12 < n) {   // (the format wants the last five bytes as literals and no match within the last twelve)
      uint32_t word;
      std::memcpy(&word, in + i, 4);
      const uint32_t slot = (word * 2654435761u) >> 20;
      const int32_t candidate = last[slot];
      last[slot] = static_cast<int32_t>(i);
      uint32_t there = 0;
      if (candidate >= 0) std::memcpy(&there, in + candidate, 4);
      if (candidate < 0 || there != word || i - candidate > 65535) { ++i; continue; }
      size_t length = 4;
      while (i + length + 5 < n && in[candidate + length] == in[i + length]) ++length;
      emit(i, i - candidate, length);
      i += length;
      anchor = i;
    }
    emit(n, 0, 0);
    return out;
  }
  template <typename T>
  static std::shared_ptr<AbstractSegment> encode_lz4(const ValueSegment<T>& segment) {
    const auto n = segment.size();
    std::vector<T> values(segment.values().begin(), segment.values().end());
    std::vector<bool> nulls(n, false);
    bool any_null = false;
    for (ChunkOffset i = 0; i < n; ++i) { nulls[i] = segment.is_null(i); any_null |= nulls[i]; if (nulls[i]) values[i] = T{}; }
    const auto* bytes = reinterpret_cast<const char*>(values.data());
    const size_t total = sizeof(T) * n, block_size = LZ4Segment<T>::BLOCK_SIZE;
    std::vector<std::vector<char>> blocks;
    for (size_t begin = 0; begin < total; begin += block_size) blocks.push_back(lz4_compress_block(bytes + begin, std::min(block_size, total - begin)));
    const size_t last_block_size = total == 0 ? 0 : (total % block_size ? total % block_size : block_size);
    return std::make_shared<LZ4Segment<T>>(std::move(blocks), any_null ? std::optional<std::vector<bool>>(std::move(nulls)) : std::nullopt, std::vector<char>{}, block_size, last_block_size, n);
  }
  // encode_all_chunks(table, SegmentEncodingSpec{type}); unsupported (type, data type) pairs stay unencoded like
  // load_and_encode_table in table_scan_test.cpp:63-75.
  static void encode_all_chunks(const std::shared_ptr<Table>& table, EncodingType type) {
    if (type == EncodingType::Unencoded) return;
    for (ChunkID chunk_id = 0; chunk_id < table->chunk_count(); ++chunk_id) {
      const auto& chunk = table->get_chunk(chunk_id);
      for (ColumnID c = 0; c < table->column_count(); ++c) {
        const auto segment = chunk->get_segment(c);
        std::shared_ptr<AbstractSegment> encoded;
        if (type == EncodingType::FrameOfReference) {
          if (const auto* ints = dynamic_cast<const ValueSegment<int32_t>*>(segment.get())) encoded = encode_frame_of_reference(*ints);
        } else if (type == EncodingType::LZ4) {   // (LZ4 string segments stay on the CPU path: strings keep their dictionary encoding here)
          if (const auto* i32 = dynamic_cast<const ValueSegment<int32_t>*>(segment.get())) encoded = encode_lz4(*i32);
          else if (const auto* i64 = dynamic_cast<const ValueSegment<int64_t>*>(segment.get())) encoded = encode_lz4(*i64);
          else if (const auto* f32 = dynamic_cast<const ValueSegment<float>*>(segment.get())) encoded = encode_lz4(*f32);
          else if (const auto* f64 = dynamic_cast<const ValueSegment<double>*>(segment.get())) encoded = encode_lz4(*f64);
          else if (const auto* str = dynamic_cast<const ValueSegment<std::string>*>(segment.get())) encoded = encode_dictionary(*str);
        } else if (const auto* i32 = dynamic_cast<const ValueSegment<int32_t>*>(segment.get())) encoded = encode_dictionary(*i32);
        else if (const auto* i64 = dynamic_cast<const ValueSegment<int64_t>*>(segment.get())) encoded = encode_dictionary(*i64);
        else if (const auto* f32 = dynamic_cast<const ValueSegment<float>*>(segment.get())) encoded = encode_dictionary(*f32);
        else if (const auto* f64 = dynamic_cast<const ValueSegment<double>*>(segment.get())) encoded = encode_dictionary(*f64);
        else if (const auto* str = dynamic_cast<const ValueSegment<std::string>*>(segment.get())) encoded = encode_dictionary(*str);
        if (encoded) chunk->replace_segment(c, encoded);
      }
    }
  }
};

// ---- marshalling into the C ABI (what INTEGRATION.md section 1 does inside Hyrise) --------------------------------------
struct DeviceColumn {
  hy_column* handle = nullptr;
  std::vector<hy_segment> descriptors;
  std::vector<std::vector<int64_t>> key_names;   // string GROUP BY columns: AggregateKeyEntry names per chunk
  ~DeviceColumn() { if (handle) hy_column_destroy(handle); }
};

// How a DictionarySegment<pmr_string> is presented to the device: the value ids alone (scans), or with a dictionary of int64
// stand-ins for the strings -- AggregateKey names (GROUP BY) or join ids (JoinHash).
enum class StringKeys { None, AggregateKeyNames, JoinIds };
struct ColumnCache {
  std::map<std::pair<ColumnID, StringKeys>, std::shared_ptr<DeviceColumn>> columns;
};
inline Table::~Table() = default;

template <typename T> constexpr uint32_t abi_type() { return static_cast<uint32_t>(data_type_of<T>()); }

// AggregateKeyEntry name of a string (aggregate_hash.cpp:852-914); strings of 5+ characters get map ids from 5e9 on.
inline int64_t string_key_name(const std::string& s, std::map<std::string, int64_t>& long_strings) {
  const auto byte = [&](size_t i) { return static_cast<int64_t>(static_cast<uint8_t>(s[i])); };
  switch (s.size()) {
    case 0: return 1;
    case 1: return 2 + byte(0);
    case 2: return 258 + (byte(1) << 8) + byte(0);
    case 3: return 65794 + (byte(2) << 16) + (byte(1) << 8) + byte(0);
    case 4: return 16843010ll + (byte(3) << 24) + (byte(2) << 16) + (byte(1) << 8) + byte(0);
    default: {
      const auto it = long_strings.emplace(s, 5000000000ll + static_cast<int64_t>(long_strings.size())).first;
      return it->second;
    }
  }
}

// JoinHash on pmr_string keys.  The reference hashes them with std::hash (join_hash_steps.hpp:282,352,573: Bloom index =
// hash % 2^20, radix partition = hash & mask) and compares them in the hash table; the device joins 64-bit integers whose
// partition and Bloom index are their low bits.  So every distinct string gets the id  unique number << 20 | hash & 0xFFFFF:
// equal ids <=> equal strings, id & mask == hash & mask for radix_bits <= 8, id % 2^20 == the Bloom index -- the join over the
// ids is the reference's join over the strings (same partitions, probe-row order, build-side insertion order).  The hash is
// libstdc++'s (_Hash_bytes, libsupc++/hash_bytes.cc, seed 0xc70f6907), spelled out so that the ids do not depend on the
// standard library this file is compiled with (same restatement as hyrise_amd/join_keys.py, pinned in tests/test_oracle_join.py).
inline uint64_t libstdcxx_hash_bytes(const void* data, size_t length) {
  const uint64_t mul = (uint64_t{0xc6a4a793} << 32) + uint64_t{0x5bd1e995};
  const auto shift_mix = [](uint64_t v) { return v ^ (v >> 47); };
  const auto* bytes = static_cast<const unsigned char*>(data);
  const size_t aligned = length & ~size_t{7};
  uint64_t hash = uint64_t{0xc70f6907} ^ (length * mul);
  for (size_t p = 0; p < aligned; p += 8) {
    uint64_t word;
    std::memcpy(&word, bytes + p, 8);
    hash ^= shift_mix(word * mul) * mul;
    hash *= mul;
  }
  if (length & 7) {
    uint64_t tail = 0;
    for (size_t n = length & 7; n-- > 0;) tail = (tail << 8) + bytes[aligned + n];
    hash ^= tail;
    hash *= mul;
  }
  hash = shift_mix(hash) * mul;
  return shift_mix(hash);
}

inline int64_t string_join_id(const std::string& s) {   // one registry for all tables: cached columns stay valid across joins
  static std::mutex mutex;
  static std::map<std::string, int64_t> ids;
  const std::lock_guard<std::mutex> lock(mutex);
  const auto [it, inserted] = ids.emplace(s, 0);
  if (inserted) it->second = static_cast<int64_t>(ids.size() << 20 | (libstdcxx_hash_bytes(s.data(), s.size()) & 0xFFFFF));
  return it->second;
}

// `string_keys`: what a string dictionary column's dictionary is replaced with (see StringKeys).
inline std::shared_ptr<DeviceColumn> device_column(const std::shared_ptr<const Table>& table, ColumnID column_id, StringKeys string_keys = StringKeys::None);

// One PosList as the `data` / `ref_chunk_id` of an HY_ENC_REFERENCE descriptor; counts which memory the lists of a column lie in.
inline void describe_pos_list(const AbstractPosList& pos_list, hy_segment& d, size_t& host_lists, size_t& device_lists) {
  if (const auto* entire = dynamic_cast<const EntireChunkPosList*>(&pos_list)) {
    d.data = nullptr; d.ref_chunk_id = entire->common_chunk_id();
  } else if (const auto* on_device = dynamic_cast<const DevicePosList*>(&pos_list)) {
    d.data = on_device->device_data();
    d.ref_chunk_id = on_device->references_single_chunk() && on_device->size() ? on_device->common_chunk_id() : 0xFFFFFFFFu;
    if (on_device->size()) ++device_lists; else d.data = nullptr, d.ref_chunk_id = 0;   // (an empty list: nothing to read in either memory)
  } else {
    const auto& rows = static_cast<const RowIDPosList&>(pos_list);
    d.data = rows.rows.data();
    d.ref_chunk_id = rows.references_single_chunk() && rows.size() ? rows.common_chunk_id() : 0xFFFFFFFFu;
    ++host_lists;
  }
}

// The chunks [chunk_begin, chunk_end) of one column on the CALLING THREAD's device (not cached: the residency cache of a table holds
// whole columns on the process's device; the shards of a DeviceGroup worker, multi_gpu.hpp, belong to that worker).
inline std::shared_ptr<DeviceColumn> device_column_of_chunks(const std::shared_ptr<const Table>& table, ColumnID column_id, StringKeys string_keys, ChunkID chunk_begin,
                                                             ChunkID chunk_end) {
  auto column = std::make_shared<DeviceColumn>();
  const auto chunk_count = chunk_end - chunk_begin;
  column->descriptors.assign(chunk_count, hy_segment{});
  column->key_names.resize(chunk_count);
  std::map<std::string, int64_t> long_strings;
  std::shared_ptr<DeviceColumn> referenced;
  size_t host_lists = 0, device_lists = 0;
  for (ChunkID table_chunk = chunk_begin; table_chunk < chunk_end; ++table_chunk) {
    const ChunkID chunk_id = table_chunk - chunk_begin;
    const auto segment = table->get_chunk(table_chunk)->get_segment(column_id);
    hy_segment& d = column->descriptors[chunk_id];
    d.size = segment->size();
    d.data_type = static_cast<uint32_t>(segment->data_type());
    d.ref_chunk_id = 0xFFFFFFFFu;
    for (const auto& sorted_by : table->get_chunk(table_chunk)->individually_sorted_by()) {   // column_vs_value_table_scan_impl.cpp:46-55
      if (sorted_by.column == column_id) d.sorted_by = static_cast<uint32_t>(sorted_by.sort_mode) + 1;   // HY_SORT_* = SortMode + 1
    }
    bool ok = false;
    const auto describe_value = [&](auto* typed) {
      using T = std::decay_t<decltype(typed->values()[0])>;
      d.encoding = HY_ENC_UNENCODED; d.width = sizeof(T); d.data = typed->values().data();
      d.nulls = typed->is_nullable() ? typed->null_words().data() : nullptr;
      ok = true;
    };
    const auto describe_lz4 = [&](auto* typed) {   // the blocks as they are: the library decompresses them on the device
      using T = std::decay_t<decltype(typed->decompress()[0])>;
      d.encoding = HY_ENC_LZ4; d.width = sizeof(T); d.data = &typed->descriptor();
      d.nulls = typed->has_nulls() ? typed->null_words().data() : nullptr;
      ok = true;
    };
    const auto describe_dictionary = [&](auto* typed) {
      d.encoding = HY_ENC_DICTIONARY; d.width = typed->attribute_vector().width; d.data = typed->attribute_vector().bytes.data();
      d.aux = typed->dictionary().data(); d.aux_size = typed->unique_values_count();
      ok = true;
    };
    if (const auto* s = dynamic_cast<const ValueSegment<int32_t>*>(segment.get())) describe_value(s);
    else if (const auto* s = dynamic_cast<const ValueSegment<int64_t>*>(segment.get())) describe_value(s);
    else if (const auto* s = dynamic_cast<const ValueSegment<float>*>(segment.get())) describe_value(s);
    else if (const auto* s = dynamic_cast<const ValueSegment<double>*>(segment.get())) describe_value(s);
    else if (const auto* s = dynamic_cast<const LZ4Segment<int32_t>*>(segment.get())) describe_lz4(s);
    else if (const auto* s = dynamic_cast<const LZ4Segment<int64_t>*>(segment.get())) describe_lz4(s);
    else if (const auto* s = dynamic_cast<const LZ4Segment<float>*>(segment.get())) describe_lz4(s);
    else if (const auto* s = dynamic_cast<const LZ4Segment<double>*>(segment.get())) describe_lz4(s);
    else if (const auto* s = dynamic_cast<const DictionarySegment<int32_t>*>(segment.get())) describe_dictionary(s);
    else if (const auto* s = dynamic_cast<const DictionarySegment<int64_t>*>(segment.get())) describe_dictionary(s);
    else if (const auto* s = dynamic_cast<const DictionarySegment<float>*>(segment.get())) describe_dictionary(s);
    else if (const auto* s = dynamic_cast<const DictionarySegment<double>*>(segment.get())) describe_dictionary(s);
    else if (const auto* s = dynamic_cast<const DictionarySegment<std::string>*>(segment.get())) {
      d.encoding = HY_ENC_DICTIONARY; d.width = s->attribute_vector().width; d.data = s->attribute_vector().bytes.data();
      d.aux_size = s->unique_values_count();
      if (string_keys != StringKeys::None) {   // GROUP BY / join key: the dictionary becomes int64 key names / join ids
        for (const auto& entry : s->dictionary())
          column->key_names[chunk_id].push_back(string_keys == StringKeys::JoinIds ? string_join_id(entry) : string_key_name(entry, long_strings));
        d.aux = column->key_names[chunk_id].data();
        d.data_type = HY_TYPE_LONG;
      } else {
        d.aux = nullptr;   // scans compare value ids; literals are resolved per chunk by the caller
      }
      ok = true;
    } else if (const auto* s = dynamic_cast<const FrameOfReferenceSegment*>(segment.get())) {
      d.encoding = HY_ENC_FRAME_OF_REFERENCE; d.width = s->offset_values().width; d.data = s->offset_values().bytes.data();
      d.aux = s->block_minima().data(); d.aux_size = static_cast<uint32_t>(s->block_minima().size());
      d.nulls = s->has_nulls() ? s->null_words().data() : nullptr;
      ok = true;
    } else if (const auto* s = dynamic_cast<const ReferenceSegment*>(segment.get())) {
      if (!referenced) referenced = device_column(s->referenced_table(), s->referenced_column_id(), string_keys);
      d.encoding = HY_ENC_REFERENCE; d.width = 8; d.ref = referenced->handle;
      if (string_keys != StringKeys::None && s->data_type() == DataType::String) d.data_type = HY_TYPE_LONG;
      describe_pos_list(*s->pos_list(), d, host_lists, device_lists);
      ok = true;
    }
    if (!ok) Fail("segment kind not handled by the device path (the Hyrise adapter keeps the stock operator here)");
  }
  // PosLists that lie in HBM are read where they are (HY_MEM_DEVICE: a reference column's only buffers are its PosLists); a table that
  // mixes them with host PosLists is handed over from the host
  if (device_lists && host_lists) {
    for (ChunkID table_chunk = chunk_begin; table_chunk < chunk_end; ++table_chunk) {
      const auto* s = dynamic_cast<const ReferenceSegment*>(table->get_chunk(table_chunk)->get_segment(column_id).get());
      if (const auto* on_device = s ? dynamic_cast<const DevicePosList*>(s->pos_list().get()) : nullptr) column->descriptors[table_chunk - chunk_begin].data = on_device->host_rows();
    }
  }
  check_status(hy_column_create(column->descriptors.data(), chunk_count, device_lists && !host_lists ? HY_MEM_DEVICE : HY_MEM_HOST, &column->handle));
  return column;
}

inline std::shared_ptr<DeviceColumn> device_column(const std::shared_ptr<const Table>& table, ColumnID column_id, StringKeys string_keys) {
  if (!table->device_columns) table->device_columns = std::make_shared<ColumnCache>();
  auto& slot = table->device_columns->columns[{column_id, string_keys}];
  if (!slot) slot = device_column_of_chunks(table, column_id, string_keys, 0, table->chunk_count());
  return slot;
}

// ---- operators ---------------------------------------------------------------------------------------------------------
class AbstractOperator {   // operators/abstract_operator.hpp
 public:
  AbstractOperator(std::shared_ptr<const AbstractOperator> left = nullptr, std::shared_ptr<const AbstractOperator> right = nullptr)
      : _left_input(std::move(left)), _right_input(std::move(right)) {}
  virtual ~AbstractOperator() = default;
  virtual const std::string& name() const = 0;
  void execute() {   // abstract_operator.cpp:78-136
    Assert(!_executed, "Operator has already been executed.");
    _output = _on_execute();
    _executed = true;
  }
  std::shared_ptr<const Table> get_output() const { Assert(_executed, "Operator has not been executed."); return _output; }
  std::shared_ptr<const Table> left_input_table() const { return _left_input->get_output(); }
  std::shared_ptr<const Table> right_input_table() const { return _right_input->get_output(); }
  void never_clear_output() {}

 protected:
  virtual std::shared_ptr<const Table> _on_execute() = 0;
  std::shared_ptr<const AbstractOperator> _left_input, _right_input;
  std::shared_ptr<const Table> _output;
  bool _executed = false;
};
using AbstractReadOnlyOperator = AbstractOperator;   // abstract_read_only_operator.hpp:18-22: ignores the transaction context

class TableWrapper : public AbstractReadOnlyOperator {   // operators/table_wrapper.hpp
 public:
  explicit TableWrapper(std::shared_ptr<const Table> table) : _table(std::move(table)) {}
  const std::string& name() const override { static const std::string n = "TableWrapper"; return n; }

 protected:
  std::shared_ptr<const Table> _on_execute() override { return _table; }
  std::shared_ptr<const Table> _table;
};

inline hy_value to_hy_value(const AllTypeVariant& v) {
  hy_value out{};
  switch (v.index()) {
    case 1: out.i32 = std::get<int32_t>(v); break;
    case 2: out.i64 = std::get<int64_t>(v); break;
    case 3: out.f32 = std::get<float>(v); break;
    case 4: out.f64 = std::get<double>(v); break;
    default: break;
  }
  return out;
}

inline bool is_between(PredicateCondition c) { return c >= PredicateCondition::BetweenInclusive && c <= PredicateCondition::BetweenExclusive; }

// The result buffers of a scan-shaped call (hy_table_scan / hy_table_scan_columns / hy_validate) in HBM: HY_SCAN_CHUNK_REGIONS -- chunk c's
// PosList starts at matches + row_base[c] and holds counts[c] RowIDs -- so the output table's PosLists are views into ONE pooled block.
// What the host needs to assemble the output table (table_scan.cpp:129-220 looks at every chunk's match count) are the counts and chunk
// states: five bytes per chunk cross the link, the RowIDs do not.
struct DeviceScanOutput {
  std::shared_ptr<DeviceBlock> matches, bookkeeping;
  std::vector<uint64_t> row_base;
  std::vector<uint32_t> counts;
  std::vector<uint8_t> states;
  hy_scan_result result{};
  uint64_t rows = 0;

  DeviceScanOutput(const Table& in_table, bool materialize_all_match) {
    const auto chunk_count = in_table.chunk_count();
    row_base.assign(size_t{chunk_count} + 1, 0);
    for (ChunkID c = 0; c < chunk_count; ++c) row_base[c + 1] = row_base[c] + in_table.get_chunk(c)->size();
    rows = row_base[chunk_count];
    counts.assign(std::max<ChunkID>(1, chunk_count), 0);
    states.assign(std::max<ChunkID>(1, chunk_count), 0);
    matches = DeviceBlock::acquire(std::max<uint64_t>(1, rows) * sizeof(RowID));
    const size_t n = std::max<ChunkID>(1, chunk_count);
    bookkeeping = DeviceBlock::acquire(8 * (n + 1) + 5 * n + 16);
    auto* base = static_cast<char*>(bookkeeping->ptr);
    result.mem = HY_MEM_DEVICE;
    result.flags = HY_SCAN_CHUNK_REGIONS | (materialize_all_match ? HY_SCAN_MATERIALIZE_ALL_MATCH : 0u);
    result.matches = static_cast<hy_row_id*>(matches->ptr);
    result.capacity = std::max<uint64_t>(1, rows);
    result.offsets = reinterpret_cast<uint64_t*>(base);
    result.counts = reinterpret_cast<uint32_t*>(base + 8 * (n + 1));
    result.chunk_state = reinterpret_cast<uint8_t*>(base + 8 * (n + 1) + 4 * n);
  }
  // after the call: counts and states of every chunk, one transfer (waits for the calling thread's stream: the PosLists are complete)
  void fetch(ChunkID chunk_count) {
    const size_t n = std::max<ChunkID>(1, chunk_count);
    std::vector<char> staged(5 * n);
    check_status(hy_memcpy_d2h(staged.data(), result.counts, 5 * n));
    std::memcpy(counts.data(), staged.data(), 4 * n);
    std::memcpy(states.data(), staged.data() + 4 * n, n);
  }
  std::shared_ptr<AbstractPosList> pos_list_of(ChunkID chunk_id) const {
    return std::make_shared<DevicePosList>(matches, result.matches + row_base[chunk_id], counts[chunk_id], true, chunk_id);
  }
};

// A scan over a reference table hands on PosLists into the DATA table: its matches are translated through the input's PosLists, once per
// distinct PosList (table_scan.cpp:158-196).  Here: once per group of columns that share their PosLists in every chunk, by ONE
// hy_poslist_translate into a pooled block (chunk regions: chunk c's translated list at block + row_base[c]).  `out` must have been
// produced with materialize_all_match.  group_of_column[c] indexes `blocks`; a null block: that group's columns are not describable to
// the device (the caller translates on the host through the lazy copies).
inline void translate_through_input_lists(const std::shared_ptr<const Table>& in_table, const DeviceScanOutput& out, std::vector<size_t>& group_of_column,
                                          std::vector<std::shared_ptr<DeviceBlock>>& blocks) {
  std::map<std::vector<const AbstractPosList*>, size_t> groups;
  group_of_column.clear();
  blocks.clear();
  for (ColumnID c = 0; c < in_table->column_count(); ++c) {
    std::vector<const AbstractPosList*> key;
    for (ChunkID k = 0; k < in_table->chunk_count(); ++k) key.push_back(std::static_pointer_cast<ReferenceSegment>(in_table->get_chunk(k)->get_segment(c))->pos_list().get());
    const auto [it, inserted] = groups.emplace(std::move(key), blocks.size());
    group_of_column.push_back(it->second);
    if (!inserted) continue;
    std::shared_ptr<DeviceBlock> block;
    try {
      const auto through = device_column(in_table, c);
      block = DeviceBlock::acquire(std::max<uint64_t>(1, out.rows) * sizeof(RowID));
      uint64_t written = 0;
      check_status(hy_poslist_translate(through->handle, &out.result, HY_POSLIST_CHUNK_REGIONS, static_cast<hy_row_id*>(block->ptr), std::max<uint64_t>(1, out.rows), &written));
    } catch (const std::logic_error&) {
      block = nullptr;
    }
    blocks.push_back(std::move(block));
  }
}

// TableScan over `column <condition> value [AND value2]` / `column IS [NOT] NULL` / `column <condition> column2`
// (the shapes create_impl maps to ColumnVsValue / ColumnBetween / ColumnIsNull / ColumnVsColumn, table_scan.cpp:312-452).
// LikeMatcher (expression/evaluation/like_matcher.hpp): `%` any run of chars, `_` one char, the rest literal; the
// *Insensitive conditions lower-case pattern and input.  Iterative two-pointer match instead of the reference's
// token/regex variants -- same language, no std::regex.
class LikeMatcher {
 public:
  LikeMatcher(std::string pattern, PredicateCondition condition) : _pattern(std::move(pattern)) {
    Assert(condition == PredicateCondition::Like || condition == PredicateCondition::NotLike || condition == PredicateCondition::LikeInsensitive ||
               condition == PredicateCondition::NotLikeInsensitive, "Expected PredicateCondition (Not)Like or (Not)LikeInsensitive.");
    _insensitive = condition == PredicateCondition::LikeInsensitive || condition == PredicateCondition::NotLikeInsensitive;
    _negated = condition == PredicateCondition::NotLike || condition == PredicateCondition::NotLikeInsensitive;
    if (_insensitive) to_lower(_pattern);
  }
  bool operator()(const std::string& input) const {
    if (!_insensitive) return matches(input) != _negated;
    auto lowered = input;
    to_lower(lowered);
    return matches(lowered) != _negated;
  }

 private:
  static void to_lower(std::string& s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c = static_cast<char>(c - 'A' + 'a'); }
  bool matches(const std::string& text) const {
    size_t t = 0, p = 0, star = std::string::npos, resume = 0;
    while (t < text.size()) {
      if (p < _pattern.size() && _pattern[p] == '%') { star = p++; resume = t; }
      else if (p < _pattern.size() && (_pattern[p] == '_' || _pattern[p] == text[t])) { ++p; ++t; }
      else if (star != std::string::npos) { p = star + 1; t = ++resume; }
      else return false;
    }
    while (p < _pattern.size() && _pattern[p] == '%') ++p;
    return p == _pattern.size();
  }
  std::string _pattern;
  bool _insensitive = false, _negated = false;
};

class TableScan : public AbstractReadOnlyOperator {
 public:
  TableScan(std::shared_ptr<const AbstractOperator> in, ColumnID column_id, PredicateCondition condition, AllTypeVariant value = NullValue{},
            std::optional<AllTypeVariant> value2 = std::nullopt)
      : AbstractReadOnlyOperator(std::move(in)), _column_id(column_id), _condition(condition), _value(std::move(value)), _value2(std::move(value2)) {}
  TableScan(std::shared_ptr<const AbstractOperator> in, ColumnID left_column, PredicateCondition condition, ColumnID right_column, bool /*column_vs_column*/)
      : AbstractReadOnlyOperator(std::move(in)), _column_id(left_column), _condition(condition), _right_column_id(right_column) {}
  const std::string& name() const override { static const std::string n = "TableScan"; return n; }
  std::vector<ChunkID> excluded_chunk_ids;
  size_t num_chunks_with_early_out = 0, num_chunks_with_all_rows_matching = 0;   // TableScan::PerformanceData (table_scan.hpp:56-69)

 protected:
  std::shared_ptr<const Table> _on_execute() override {
    const auto in_table = left_input_table();
    const auto chunk_count = in_table->chunk_count();
    const auto column = device_column(in_table, _column_id);
    // Device-resident results (the default): the PosLists stay in a pooled block of HBM, the output table's ReferenceSegments hold
    // DevicePosLists into it.  Host results: the library copies the RowIDs back (the boundary as rounds 1-5 used it).
    std::unique_ptr<DeviceScanOutput> on_device;
    if (device_resident_results()) on_device = std::make_unique<DeviceScanOutput>(*in_table, in_table->type() == TableType::References);
    std::vector<RowID> matches(on_device ? 1 : std::max<uint64_t>(1, in_table->row_count()));
    std::vector<uint64_t> offsets(chunk_count + 1);
    std::vector<uint32_t> counts(std::max<ChunkID>(1, chunk_count));
    std::vector<uint8_t> states(std::max<ChunkID>(1, chunk_count));
    hy_scan_result result{};
    result.mem = HY_MEM_HOST;
    result.matches = reinterpret_cast<hy_row_id*>(matches.data());
    result.capacity = in_table->row_count();
    result.offsets = offsets.data();
    result.counts = counts.data();
    result.chunk_state = states.data();
    if (on_device) result = on_device->result;
    bool evaluated_on_host = false;
    if (_right_column_id) {
      check_status(hy_table_scan_columns(column->handle, device_column(in_table, *_right_column_id)->handle, static_cast<uint32_t>(_condition), &result));
    } else {
      hy_predicate predicate{};
      predicate.condition = static_cast<uint32_t>(_condition);
      predicate.column_is_nullable = in_table->column_is_nullable(_column_id);
      const bool null_test = _condition == PredicateCondition::IsNull || _condition == PredicateCondition::IsNotNull;
      std::vector<uint32_t> lower, upper;
      std::vector<uint8_t> found;
      const bool like = _condition == PredicateCondition::Like || _condition == PredicateCondition::NotLike ||
                        _condition == PredicateCondition::LikeInsensitive || _condition == PredicateCondition::NotLikeInsensitive;
      std::vector<uint64_t> match_words, match_word_offsets;
      if (like) {
        // ColumnLikeTableScanImpl (column_like_table_scan_impl.cpp:28-36)
        Assert(in_table->column_data_type(_column_id) == DataType::String, "LIKE operator only applicable on string columns.");
        Assert(std::holds_alternative<std::string>(_value), "Right parameter must be a string.");
        find_matches_in_dictionaries(in_table, match_words, match_word_offsets, predicate);
      } else if (!null_test) {
        const auto column_type = in_table->column_data_type(_column_id);
        if (column_type == DataType::String) {
          // ColumnVsValueTableScanImpl asserts matching types (column_vs_value_table_scan_impl.cpp:34-36)
          Assert(data_type_from_all_type_variant(_value) == DataType::String, "Cannot scan: column and value data type do not match.");
          predicate.value_type = static_cast<uint32_t>(DataType::String);
          resolve_string_literal(in_table, lower, upper, found, predicate);
        } else {
          // TableScan::create_impl casts the literal(s) to the column's type, adjusting the condition where only that is
          // lossless (table_scan.cpp:336-366, 406-448); where neither works the stock ExpressionEvaluator scan runs.
          const hy_value first = to_hy_value(_value), second = _value2 ? to_hy_value(*_value2) : hy_value{};
          hy_predicate cast{};
          const auto status = hy_predicate_cast(static_cast<uint32_t>(_condition), static_cast<uint32_t>(column_type), static_cast<uint32_t>(data_type_from_all_type_variant(_value)),
                                                &first, _value2 ? static_cast<uint32_t>(data_type_from_all_type_variant(*_value2)) : HY_TYPE_NULL, _value2 ? &second : nullptr, &cast);
          const auto numeric_literal = [](const AllTypeVariant& v) { return v.index() >= 1 && v.index() <= 4; };
          // a string literal against a numeric column is an invalid plan, not a scan for the evaluator (table_scan_test.cpp:383-405: std::logic_error)
          Assert(status != HY_ERR_UNSUPPORTED || (numeric_literal(_value) && (!_value2 || numeric_literal(*_value2))), "Cannot scan: column and value data type do not match.");
          if (status == HY_ERR_UNSUPPORTED) {
            // No lossless cast (`float_column = 3.1`, `int_column < 3.5`, a literal outside the column type's range): the reference falls
            // back to its ExpressionEvaluator scan (table_scan.cpp:346-366, 450), and so does the adapter -- the stock operator runs.
            // The mirror's stand-in for it: every row compared with the literal in the common type (expression_functors.hpp).
            on_device.reset();
            matches.resize(std::max<uint64_t>(1, in_table->row_count()));
            evaluate_on_host(in_table, matches, offsets, counts, states);
            evaluated_on_host = true;
          } else {
            check_status(status);
            predicate.condition = cast.condition, predicate.value_type = cast.value_type, predicate.value = cast.value, predicate.value2 = cast.value2;
          }
        }
      }
      if (!evaluated_on_host) check_status(hy_table_scan(column->handle, &predicate, excluded_chunk_ids.data(), static_cast<uint32_t>(excluded_chunk_ids.size()), &result));
    }
    // ---- output assembly, table_scan.cpp:129-220 ----
    std::vector<size_t> group_of_column;
    std::vector<std::shared_ptr<DeviceBlock>> translated;
    const RowID* device_matches_on_host = nullptr;   // (only for column groups the device could not translate)
    if (on_device) {
      on_device->fetch(chunk_count);
      counts = on_device->counts;
      states = on_device->states;
      offsets = on_device->row_base;   // (chunk regions)
      bool any_partial = false;
      for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) any_partial = any_partial || (counts[chunk_id] && counts[chunk_id] != in_table->get_chunk(chunk_id)->size());
      if (in_table->type() == TableType::References && any_partial) {
        translate_through_input_lists(in_table, *on_device, group_of_column, translated);
        if (std::any_of(translated.begin(), translated.end(), [](const auto& b) { return !b; })) {
          on_device->matches->prefetch_to_host(on_device->rows);
          device_matches_on_host = on_device->matches->host_copy();
        }
      }
    }
    std::vector<std::shared_ptr<Chunk>> output_chunks;
    for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) {
      if (states[chunk_id] == HY_CHUNK_NONE_MATCH) ++num_chunks_with_early_out;
      if (states[chunk_id] == HY_CHUNK_ALL_MATCH) ++num_chunks_with_all_rows_matching;
      if (counts[chunk_id] == 0) continue;   // :132-134
      const auto chunk_in = in_table->get_chunk(chunk_id);
      const bool all = counts[chunk_id] == chunk_in->size();
      Segments out_segments;
      if (in_table->type() == TableType::References) {
        if (all) {   // forward the chunk (:151-156)
          for (ColumnID c = 0; c < in_table->column_count(); ++c) out_segments.push_back(chunk_in->get_segment(c));
        } else {     // translate through every distinct input pos list once (:158-196)
          std::map<const AbstractPosList*, std::shared_ptr<AbstractPosList>> filtered;
          for (ColumnID c = 0; c < in_table->column_count(); ++c) {
            const auto ref = std::static_pointer_cast<ReferenceSegment>(chunk_in->get_segment(c));
            auto& list = filtered[ref->pos_list().get()];
            if (!list && on_device && translated[group_of_column[c]]) {   // already translated, in HBM
              const auto& block = translated[group_of_column[c]];
              list = std::make_shared<DevicePosList>(block, static_cast<const hy_row_id*>(block->ptr) + offsets[chunk_id], counts[chunk_id], ref->pos_list()->references_single_chunk(),
                                                     ref->pos_list()->common_chunk_id());
            } else if (!list) {
              const RowID* chunk_matches = on_device ? device_matches_on_host + offsets[chunk_id] : matches.data() + offsets[chunk_id];
              auto rows = std::make_shared<RowIDPosList>();
              for (uint64_t m = 0; m < counts[chunk_id]; ++m) rows->rows.push_back((*ref->pos_list())[chunk_matches[m].chunk_offset]);
              if (ref->pos_list()->references_single_chunk()) rows->guarantee_single_chunk();
              list = rows;
            }
            out_segments.push_back(std::make_shared<ReferenceSegment>(ref->referenced_table(), ref->referenced_column_id(), list));
          }
        }
      } else {
        std::shared_ptr<AbstractPosList> pos_list;
        if (all) pos_list = std::make_shared<EntireChunkPosList>(chunk_id, chunk_in->size());   // :201-205
        else if (on_device) pos_list = on_device->pos_list_of(chunk_id);
        else {
          auto rows = std::make_shared<RowIDPosList>(std::vector<RowID>(matches.begin() + offsets[chunk_id], matches.begin() + offsets[chunk_id + 1]));
          rows->guarantee_single_chunk();
          pos_list = rows;
        }
        for (ColumnID c = 0; c < in_table->column_count(); ++c) out_segments.push_back(std::make_shared<ReferenceSegment>(in_table, c, pos_list));
      }
      output_chunks.push_back(std::make_shared<Chunk>(std::move(out_segments)));
    }
    return std::make_shared<Table>(in_table->column_definitions(), TableType::References, std::move(output_chunks));
  }

 private:
  // The stock scan's stand-in (see _on_execute): a numeric column against numeric literals, row by row, both sides in the common type
  // (two integers: int64; otherwise double -- what the ExpressionEvaluator's comparison functors promote to).  NULL matches nothing.
  void evaluate_on_host(const std::shared_ptr<const Table>& in_table, std::vector<RowID>& matches, std::vector<uint64_t>& offsets, std::vector<uint32_t>& counts,
                        std::vector<uint8_t>& states) const {
    const auto as_double = [](const AllTypeVariant& v) {
      return std::visit([](const auto& x) -> double { if constexpr (std::is_arithmetic_v<std::decay_t<decltype(x)>>) return static_cast<double>(x); else return 0.0; }, v);
    };
    const auto integral = [](const AllTypeVariant& v) { return v.index() == 1 || v.index() == 2; };
    const auto as_long = [](const AllTypeVariant& v) { return v.index() == 1 ? static_cast<int64_t>(std::get<int32_t>(v)) : std::get<int64_t>(v); };
    const auto compare = [&](const AllTypeVariant& cell, const AllTypeVariant& literal) {   // -1, 0, 1
      if (integral(cell) && integral(literal)) { const int64_t a = as_long(cell), b = as_long(literal); return a < b ? -1 : a > b ? 1 : 0; }
      const double a = as_double(cell), b = as_double(literal);
      return a < b ? -1 : a > b ? 1 : 0;
    };
    const auto matches_row = [&](const AllTypeVariant& cell) {
      if (variant_is_null(cell)) return false;
      const int first = compare(cell, _value);
      switch (_condition) {
        case PredicateCondition::Equals: return first == 0;
        case PredicateCondition::NotEquals: return first != 0;
        case PredicateCondition::LessThan: return first < 0;
        case PredicateCondition::LessThanEquals: return first <= 0;
        case PredicateCondition::GreaterThan: return first > 0;
        case PredicateCondition::GreaterThanEquals: return first >= 0;
        default: break;
      }
      Assert(_value2.has_value(), "BETWEEN needs two literals.");
      const int second = compare(cell, *_value2);
      const bool lower_inclusive = _condition == PredicateCondition::BetweenInclusive || _condition == PredicateCondition::BetweenUpperExclusive;
      const bool upper_inclusive = _condition == PredicateCondition::BetweenInclusive || _condition == PredicateCondition::BetweenLowerExclusive;
      return (lower_inclusive ? first >= 0 : first > 0) && (upper_inclusive ? second <= 0 : second < 0);
    };
    uint64_t written = 0;
    for (ChunkID chunk_id = 0; chunk_id < in_table->chunk_count(); ++chunk_id) {
      const auto segment = in_table->get_chunk(chunk_id)->get_segment(_column_id);
      offsets[chunk_id] = written;
      const bool excluded = std::find(excluded_chunk_ids.begin(), excluded_chunk_ids.end(), chunk_id) != excluded_chunk_ids.end();
      for (ChunkOffset offset = 0; !excluded && offset < segment->size(); ++offset) {
        if (matches_row((*segment)[offset])) matches[written++] = RowID{chunk_id, offset};
      }
      counts[chunk_id] = static_cast<uint32_t>(written - offsets[chunk_id]);
      states[chunk_id] = HY_CHUNK_SCANNED;
    }
    offsets[in_table->chunk_count()] = written;
  }

  // DictionarySegment<pmr_string>: resolve the literal per chunk like _get_search_value_id
  // (column_vs_value_table_scan_impl.cpp:211-226) and column_between_table_scan_impl.cpp:112-124.
  void resolve_string_literal(const std::shared_ptr<const Table>& in_table, std::vector<uint32_t>& lower, std::vector<uint32_t>& upper,
                              std::vector<uint8_t>& found, hy_predicate& predicate) const {
    auto data_table = in_table;
    auto column_id = _column_id;
    if (in_table->type() == TableType::References && in_table->chunk_count()) {
      const auto ref = std::static_pointer_cast<ReferenceSegment>(in_table->get_chunk(0)->get_segment(_column_id));
      data_table = ref->referenced_table();
      column_id = ref->referenced_column_id();
    }
    const auto& value = std::get<std::string>(_value);
    for (ChunkID c = 0; c < data_table->chunk_count(); ++c) {
      const auto* dict = dynamic_cast<const DictionarySegment<std::string>*>(data_table->get_chunk(c)->get_segment(column_id).get());
      Assert(dict, "unencoded string segments stay on the CPU path");
      if (is_between(_condition)) {
        const auto& value2 = std::get<std::string>(*_value2);
        const bool lower_inclusive = _condition == PredicateCondition::BetweenInclusive || _condition == PredicateCondition::BetweenUpperExclusive;
        const bool upper_inclusive = _condition == PredicateCondition::BetweenInclusive || _condition == PredicateCondition::BetweenLowerExclusive;
        lower.push_back(lower_inclusive ? dict->lower_bound(value) : dict->upper_bound(value));
        upper.push_back(upper_inclusive ? dict->upper_bound(value2) : dict->lower_bound(value2));
        found.push_back(0);
      } else {
        const auto lb = dict->lower_bound(value);
        lower.push_back(lb);
        upper.push_back(dict->upper_bound(value));
        found.push_back(lb != HY_INVALID_VALUE_ID && dict->dictionary()[lb] == value);
      }
    }
    predicate.value_type = HY_TYPE_STRING;
    predicate.per_chunk_lower = lower.data();
    predicate.per_chunk_upper = upper.data();
    predicate.per_chunk_found = found.data();
  }

  // _find_matches_in_dictionary per data chunk (column_like_table_scan_impl.cpp:142-159): bit i of chunk c's words says
  // whether dictionary entry i satisfies the (possibly negated) pattern; the device tests value ids against it.
  void find_matches_in_dictionaries(const std::shared_ptr<const Table>& in_table, std::vector<uint64_t>& words, std::vector<uint64_t>& word_offsets,
                                    hy_predicate& predicate) const {
    auto data_table = in_table;
    auto column_id = _column_id;
    if (in_table->type() == TableType::References && in_table->chunk_count()) {
      const auto ref = std::static_pointer_cast<ReferenceSegment>(in_table->get_chunk(0)->get_segment(_column_id));
      data_table = ref->referenced_table();
      column_id = ref->referenced_column_id();
    }
    const LikeMatcher matcher(std::get<std::string>(_value), _condition);
    word_offsets.push_back(0);
    for (ChunkID c = 0; c < data_table->chunk_count(); ++c) {
      const auto* dict = dynamic_cast<const DictionarySegment<std::string>*>(data_table->get_chunk(c)->get_segment(column_id).get());
      Assert(dict, "unencoded string segments stay on the CPU path");
      const auto& dictionary = dict->dictionary();
      const size_t base = words.size();
      words.resize(base + (dictionary.size() + 63) / 64, 0);
      for (size_t i = 0; i < dictionary.size(); ++i) if (matcher(dictionary[i])) words[base + i / 64] |= uint64_t{1} << (i % 64);
      word_offsets.push_back(words.size());
    }
    if (words.empty()) words.push_back(0);
    predicate.value_type = HY_TYPE_STRING;
    predicate.match_words = words.data();
    predicate.match_word_offsets = word_offsets.data();
  }

  ColumnID _column_id;
  PredicateCondition _condition;
  AllTypeVariant _value;
  std::optional<AllTypeVariant> _value2;
  std::optional<ColumnID> _right_column_id;
};

// ---- Validate (operators/validate.hpp, validate.cpp:87-314) -------------------------------------------------------------
struct TransactionContext {   // concurrency/transaction_context.hpp: what Validate reads of it
  TransactionContext(TransactionID transaction_id, CommitID snapshot_commit_id, bool has_in_flight_delete = false)
      : _tid(transaction_id), _snapshot(snapshot_commit_id), _delete(has_in_flight_delete) {}
  TransactionID transaction_id() const { return _tid; }
  CommitID snapshot_commit_id() const { return _snapshot; }
  bool has_in_flight_delete() const { return _delete; }   // read_write_operators() contains a Delete (validate.cpp:116-127)

 private:
  TransactionID _tid;
  CommitID _snapshot;
  bool _delete;
};

// The MvccData of a data table as a column of HY_ENC_MVCC segments.  Mutable chunks change under writers, so this column is
// rebuilt per call (the Hyrise adapter caches the immutable chunks).
inline std::shared_ptr<DeviceColumn> mvcc_column(const std::shared_ptr<const Table>& table) {
  auto column = std::make_shared<DeviceColumn>();
  const auto chunk_count = table->chunk_count();
  column->descriptors.assign(chunk_count, hy_segment{});
  for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) {
    const auto& chunk = table->get_chunk(chunk_id);
    Assert(chunk->has_mvcc_data(), "Trying to use Validate on a table that has no MVCC data.");
    const auto& mvcc = *chunk->mvcc_data();
    hy_segment& d = column->descriptors[chunk_id];
    d.encoding = HY_ENC_MVCC; d.data_type = HY_TYPE_INT; d.size = chunk->size(); d.width = 4;
    d.data = mvcc.tids.data(); d.aux = mvcc.begin_cids.data(); d.nulls = reinterpret_cast<const uint64_t*>(mvcc.end_cids.data());
    d.aux_size = mvcc.max_begin_cid;
    d.ref_chunk_id = chunk->invalid_row_count() | (chunk->is_mutable() ? 1u << 31 : 0u);
  }
  check_status(hy_column_create(column->descriptors.data(), chunk_count, HY_MEM_HOST, &column->handle));
  return column;
}

class Validate : public AbstractReadOnlyOperator {
 public:
  explicit Validate(std::shared_ptr<const AbstractOperator> in) : AbstractReadOnlyOperator(std::move(in)) {}
  const std::string& name() const override { static const std::string n = "Validate"; return n; }
  void set_transaction_context(std::shared_ptr<TransactionContext> context) { _context = std::move(context); }
  static bool is_row_visible(TransactionID our_tid, CommitID snapshot_commit_id, TransactionID row_tid, CommitID begin_cid, CommitID end_cid) {
    return snapshot_commit_id < end_cid && ((snapshot_commit_id >= begin_cid) != (row_tid == our_tid));   // validate.cpp:47-55
  }

 protected:
  std::shared_ptr<const Table> _on_execute() override {
    if (!_context) Fail("Validate cannot be called without a transaction context.");   // validate.cpp:87-89
    const auto in_table = left_input_table();
    const auto chunk_count = in_table->chunk_count();
    // the table that owns the MvccData: the input itself or the table its reference segments point to (:186-199)
    auto mvcc_table = in_table;
    if (in_table->type() == TableType::References && chunk_count) {
      mvcc_table = std::static_pointer_cast<ReferenceSegment>(in_table->get_chunk(0)->get_segment(0))->referenced_table();
    }
    const auto mvcc = mvcc_column(mvcc_table);
    std::shared_ptr<DeviceColumn> input = mvcc;
    if (in_table->type() == TableType::References) {   // the input's pos lists over the MVCC column
      input = std::make_shared<DeviceColumn>();
      input->descriptors.assign(chunk_count, hy_segment{});
      size_t host_lists = 0, device_lists = 0;
      for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) {
        const auto ref = std::static_pointer_cast<ReferenceSegment>(in_table->get_chunk(chunk_id)->get_segment(0));
        hy_segment& d = input->descriptors[chunk_id];
        d.encoding = HY_ENC_REFERENCE; d.data_type = HY_TYPE_INT; d.size = ref->size(); d.width = 8; d.ref = mvcc->handle;
        describe_pos_list(*ref->pos_list(), d, host_lists, device_lists);
      }
      if (device_lists && host_lists) {   // (a mixed table is handed over from the host, see device_column_of_chunks)
        for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) {
          const auto ref = std::static_pointer_cast<ReferenceSegment>(in_table->get_chunk(chunk_id)->get_segment(0));
          if (const auto* on_device = dynamic_cast<const DevicePosList*>(ref->pos_list().get())) input->descriptors[chunk_id].data = on_device->host_rows();
        }
      }
      check_status(hy_column_create(input->descriptors.data(), chunk_count, device_lists && !host_lists ? HY_MEM_DEVICE : HY_MEM_HOST, &input->handle));
    }
    std::unique_ptr<DeviceScanOutput> on_device;
    if (device_resident_results()) on_device = std::make_unique<DeviceScanOutput>(*in_table, in_table->type() == TableType::References);
    std::vector<RowID> matches(on_device ? 1 : std::max<uint64_t>(1, in_table->row_count()));
    std::vector<uint64_t> offsets(chunk_count + 1);
    std::vector<uint32_t> counts(std::max<ChunkID>(1, chunk_count));
    std::vector<uint8_t> states(std::max<ChunkID>(1, chunk_count));
    hy_scan_result result{};
    result.mem = HY_MEM_HOST;
    result.matches = reinterpret_cast<hy_row_id*>(matches.data());
    result.capacity = in_table->row_count();
    result.offsets = offsets.data();
    result.counts = counts.data();
    result.chunk_state = states.data();
    if (on_device) result = on_device->result;
    check_status(hy_validate(input->handle, _context->transaction_id(), _context->snapshot_commit_id(), _context->has_in_flight_delete() ? 0 : 1, &result));
    std::shared_ptr<DeviceBlock> translated;   // reference input: the visible rows' RowIDs in the data table, chunk regions
    if (on_device) {
      on_device->fetch(chunk_count);
      counts = on_device->counts;
      states = on_device->states;
      offsets = on_device->row_base;
      bool any_partial = false;
      for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) any_partial = any_partial || (counts[chunk_id] && states[chunk_id] != HY_CHUNK_ALL_MATCH);
      if (in_table->type() == TableType::References && any_partial) {
        translated = DeviceBlock::acquire(std::max<uint64_t>(1, on_device->rows) * sizeof(RowID));
        uint64_t written = 0;
        check_status(hy_poslist_translate(input->handle, &on_device->result, HY_POSLIST_CHUNK_REGIONS, static_cast<hy_row_id*>(translated->ptr), std::max<uint64_t>(1, on_device->rows), &written));
      }
    }
    // ---- output assembly, validate.cpp:256-311 ----
    std::vector<std::shared_ptr<Chunk>> output_chunks;
    for (ChunkID chunk_id = 0; chunk_id < chunk_count; ++chunk_id) {
      if (counts[chunk_id] == 0) continue;   // :299
      const auto chunk_in = in_table->get_chunk(chunk_id);
      const bool entirely_visible = states[chunk_id] == HY_CHUNK_ALL_MATCH;
      Segments out_segments;
      if (in_table->type() == TableType::References) {
        const auto first = std::static_pointer_cast<ReferenceSegment>(chunk_in->get_segment(0));
        std::shared_ptr<const AbstractPosList> pos_list = first->pos_list();   // reused when entirely visible (:207-211)
        if (!entirely_visible && on_device) {
          pos_list = std::make_shared<DevicePosList>(translated, static_cast<const hy_row_id*>(translated->ptr) + offsets[chunk_id], counts[chunk_id],
                                                     first->pos_list()->references_single_chunk(), first->pos_list()->common_chunk_id());
        } else if (!entirely_visible) {
          auto visible = std::make_shared<RowIDPosList>();
          for (uint64_t m = offsets[chunk_id]; m < offsets[chunk_id + 1]; ++m) visible->rows.push_back((*first->pos_list())[matches[m].chunk_offset]);
          if (first->pos_list()->references_single_chunk()) visible->guarantee_single_chunk();
          pos_list = visible;
        }
        for (ColumnID c = 0; c < in_table->column_count(); ++c) {
          const auto ref = std::static_pointer_cast<ReferenceSegment>(chunk_in->get_segment(c));
          out_segments.push_back(std::make_shared<ReferenceSegment>(ref->referenced_table(), ref->referenced_column_id(), pos_list));
        }
      } else {
        std::shared_ptr<AbstractPosList> pos_list;
        if (entirely_visible) pos_list = std::make_shared<EntireChunkPosList>(chunk_id, chunk_in->size());   // :282-284
        else if (on_device) pos_list = on_device->pos_list_of(chunk_id);
        else {
          auto rows = std::make_shared<RowIDPosList>(std::vector<RowID>(matches.begin() + offsets[chunk_id], matches.begin() + offsets[chunk_id + 1]));
          rows->guarantee_single_chunk();
          pos_list = rows;
        }
        for (ColumnID c = 0; c < in_table->column_count(); ++c) out_segments.push_back(std::make_shared<ReferenceSegment>(in_table, c, pos_list));
      }
      output_chunks.push_back(std::make_shared<Chunk>(std::move(out_segments)));
    }
    return std::make_shared<Table>(in_table->column_definitions(), TableType::References, std::move(output_chunks));
  }

 private:
  std::shared_ptr<TransactionContext> _context;
};

using ColumnIDPair = std::pair<ColumnID, ColumnID>;

struct OperatorJoinPredicate {   // operators/operator_join_predicate.hpp: left column <condition> right column
  ColumnIDPair column_ids;
  PredicateCondition predicate_condition;
};

class JoinHash : public AbstractReadOnlyOperator {   // operators/join_hash.hpp:33-36 (primary predicate: Equals)
 public:
  JoinHash(std::shared_ptr<const AbstractOperator> left, std::shared_ptr<const AbstractOperator> right, JoinMode mode, ColumnIDPair column_ids,
           std::optional<size_t> radix_bits = std::nullopt, std::vector<OperatorJoinPredicate> secondary_predicates = {})
      : AbstractReadOnlyOperator(std::move(left), std::move(right)), _mode(mode), _column_ids(column_ids), _radix_bits(radix_bits),
        _secondary_predicates(std::move(secondary_predicates)) {}
  const std::string& name() const override { static const std::string n = "JoinHash"; return n; }
  size_t radix_bits = 0;
  bool left_input_is_build_side = false;   // JoinHash::PerformanceData

 protected:
  std::shared_ptr<const Table> _on_execute() override {
    const auto left = left_input_table(), right = right_input_table();
    // string keys join as ids (see string_join_id); a string against a number stays with the stock operator
    const bool string_keys = left->column_data_type(_column_ids.first) == DataType::String;
    Assert(string_keys == (right->column_data_type(_column_ids.second) == DataType::String), "JoinHash: a string key against a numeric key is not run on the device");
    const auto keys = string_keys ? StringKeys::JoinIds : StringKeys::None;
    const auto left_column = device_column(left, _column_ids.first, keys), right_column = device_column(right, _column_ids.second, keys);
    // JoinHash::supports (join_hash.cpp:39-44)
    Assert(_mode != JoinMode::AntiNullAsTrue || _secondary_predicates.empty(), "JoinHash does not support secondary predicates with AntiNullAsTrue");
    std::vector<hy_join_predicate> secondary;
    std::vector<std::shared_ptr<DeviceColumn>> secondary_columns;   // (keeps the handles alive)
    for (const auto& predicate : _secondary_predicates) {
      const auto l = device_column(left, predicate.column_ids.first), r = device_column(right, predicate.column_ids.second);
      secondary_columns.push_back(l);
      secondary_columns.push_back(r);
      secondary.push_back(hy_join_predicate{l->handle, r->handle, static_cast<uint32_t>(predicate.predicate_condition), 0});
    }
    // One call: room for one partner per row of the larger input (every key / foreign-key join fits); a join that multiplies
    // rows reports what it needs with HY_ERR_CAPACITY (nothing written) and runs once more with exactly that.
    uint64_t capacity = std::max<uint64_t>(1, std::max(left->row_count(), right->row_count()));
    uint32_t slice_capacity = static_cast<uint32_t>(capacity / 131070 + std::max(left->chunk_count(), right->chunk_count()) + 300);
    const bool semi_anti = _mode == JoinMode::Semi || _mode == JoinMode::AntiNullAsTrue || _mode == JoinMode::AntiNullAsFalse;
    const bool on_device = device_resident_results();
    std::vector<RowID> left_positions, right_positions;
    std::vector<uint64_t> slice_offsets;
    // Device-resident results: both PosLists in blocks of the library's result-buffer pool (the pair its calibration prefers, the second
    // list 1.25 MiB past the 2 MiB grid the first starts on); what the host reads to cut the output into chunks are the PosList
    // boundaries -- eight bytes per 131 070 pairs.
    std::shared_ptr<DeviceBlock> left_block, right_block, offsets_block;
    hy_join_result result{};
    for (int attempt = 0;; ++attempt) {
      slice_offsets.assign(size_t{slice_capacity} + 2, 0);
      result = hy_join_result{};
      result.radix_bits = _radix_bits ? static_cast<uint32_t>(*_radix_bits) : 0xFFFFFFFFu;
      result.capacity = capacity;
      result.slice_capacity = slice_capacity;
      if (on_device) {
        left_block.reset(); right_block.reset();   // (a second attempt: the first one's blocks go back first)
        hy_row_id* l = nullptr;
        hy_row_id* r = nullptr;
        check_status(hy_result_pool_acquire_pair(capacity, &l, &r));
        left_block = std::make_shared<DeviceBlock>(l);
        right_block = std::make_shared<DeviceBlock>(r);
        offsets_block = DeviceBlock::acquire(8 * (size_t{slice_capacity} + 2));
        result.mem = HY_MEM_DEVICE;
        result.left_pos = l;
        result.right_pos = semi_anti ? l : r;
        result.slice_offsets = static_cast<uint64_t*>(offsets_block->ptr);
      } else {
        left_positions.resize(capacity);
        right_positions.resize(capacity);
        result.mem = HY_MEM_HOST;
        result.left_pos = reinterpret_cast<hy_row_id*>(left_positions.data());
        result.right_pos = reinterpret_cast<hy_row_id*>(right_positions.data());
        result.slice_offsets = slice_offsets.data();
      }
      const auto status = hy_join_hash_predicates(left_column->handle, right_column->handle, static_cast<uint32_t>(_mode), secondary.data(), static_cast<uint32_t>(secondary.size()), &result);
      if (status == HY_ERR_CAPACITY && attempt == 0 && (result.n_pairs > capacity || result.n_slices > slice_capacity)) {
        capacity = std::max<uint64_t>(capacity, result.n_pairs);
        slice_capacity = std::max(slice_capacity, result.n_slices);
        continue;
      }
      check_status(status);
      break;
    }
    if (on_device) {
      if (semi_anti) right_block.reset();
      check_status(hy_memcpy_d2h(slice_offsets.data(), result.slice_offsets, 8 * (size_t{result.n_slices} + 1)));
    }
    radix_bits = result.radix_bits;
    left_input_is_build_side = result.left_is_build;
    // Output (write_output_chunks, join_output_writing.cpp:205-340): one chunk per non-empty PosList, small ones merged by the
    // reference's 1000 / 4000 rule (hy_join_output_chunks).  Columns: left input's, then right input's (Semi/Anti: left only).
    TableColumnDefinitions definitions = left->column_definitions();
    if (!semi_anti) for (const auto& d : right->column_definitions()) definitions.push_back({d.name, d.data_type, d.nullable || _mode == JoinMode::Left});
    if (_mode == JoinMode::Right) for (ColumnID c = 0; c < left->column_count(); ++c) definitions[c].nullable = true;
    std::vector<uint64_t> chunk_offsets(size_t{result.n_slices} + 1);
    uint32_t n_output_chunks = 0;
    check_status(hy_join_output_chunks(slice_offsets.data(), result.n_slices, chunk_offsets.data(), &n_output_chunks));
    std::vector<std::shared_ptr<Chunk>> chunks;
    if (on_device) {
      DeviceSide left_side(left, left_block, result.n_pairs), right_side(right, semi_anti ? nullptr : right_block, semi_anti ? 0 : result.n_pairs);
      for (uint32_t k = 0; k < n_output_chunks; ++k) {
        Segments segments;
        left_side.append(segments, chunk_offsets[k], chunk_offsets[k + 1]);
        if (!semi_anti) right_side.append(segments, chunk_offsets[k], chunk_offsets[k + 1]);
        chunks.push_back(std::make_shared<Chunk>(std::move(segments)));
      }
      return std::make_shared<Table>(definitions, TableType::References, std::move(chunks));
    }
    for (uint32_t k = 0; k < n_output_chunks; ++k) {
      const auto begin = chunk_offsets[k], end = chunk_offsets[k + 1];
      Segments segments;
      append_side(segments, left, std::vector<RowID>(left_positions.begin() + begin, left_positions.begin() + end));
      if (!semi_anti) append_side(segments, right, std::vector<RowID>(right_positions.begin() + begin, right_positions.begin() + end));
      chunks.push_back(std::make_shared<Chunk>(std::move(segments)));
    }
    return std::make_shared<Table>(definitions, TableType::References, std::move(chunks));
  }

 private:
  // write_output_segments (join_output_writing.cpp:95-200) over PosLists in HBM: one side of the join result.  A data input's positions
  // ARE its RowIDs -- every output chunk's PosList is a view into the join's block.  A reference input's positions are dereferenced through
  // the input's PosLists (once per group of columns that share them: hy_poslist_gather over the WHOLE pair array into another pooled
  // block); a group the device cannot describe is dereferenced on the host through the lazy copies.
  struct DeviceSide {
    DeviceSide(const std::shared_ptr<const Table>& input, std::shared_ptr<DeviceBlock> positions, uint64_t n_pairs) : _input(input), _positions(std::move(positions)), _n(n_pairs) {
      if (!_positions || input->type() == TableType::Data) return;
      std::map<std::vector<const AbstractPosList*>, size_t> groups;
      for (ColumnID c = 0; c < input->column_count(); ++c) {
        std::vector<const AbstractPosList*> key;
        for (ChunkID k = 0; k < input->chunk_count(); ++k) key.push_back(std::static_pointer_cast<ReferenceSegment>(input->get_chunk(k)->get_segment(c))->pos_list().get());
        const auto [it, inserted] = groups.emplace(key, _resolved.size());
        _group_of_column.push_back(it->second);
        if (!inserted) continue;
        std::shared_ptr<DeviceBlock> resolved;
        if (_n) {
          try {
            const auto through = device_column(input, c);
            resolved = DeviceBlock::acquire(_n * sizeof(RowID));
            check_status(hy_poslist_gather(through->handle, static_cast<const hy_row_id*>(_positions->ptr), _n, static_cast<hy_row_id*>(resolved->ptr)));
          } catch (const std::logic_error&) {
            resolved = nullptr;
          }
        }
        _resolved.push_back(std::move(resolved));
        _lists.push_back(std::move(key));
      }
      if (_n) check_status(hy_synchronize());   // (the gathered lists are complete before another thread's operator may read them)
    }
    void append(Segments& segments, uint64_t begin, uint64_t end) {
      if (_input->type() == TableType::Data) {
        const auto pos_list = std::make_shared<DevicePosList>(_positions, static_cast<const hy_row_id*>(_positions->ptr) + begin, end - begin);
        for (ColumnID c = 0; c < _input->column_count(); ++c) segments.push_back(std::make_shared<ReferenceSegment>(_input, c, pos_list));
        return;
      }
      std::vector<std::shared_ptr<AbstractPosList>> of_group(_resolved.size());
      for (ColumnID c = 0; c < _input->column_count(); ++c) {
        const size_t g = _group_of_column[c];
        if (!of_group[g] && _resolved[g]) {
          of_group[g] = std::make_shared<DevicePosList>(_resolved[g], static_cast<const hy_row_id*>(_resolved[g]->ptr) + begin, end - begin);
        } else if (!of_group[g]) {
          _positions->prefetch_to_host(_n);
          const RowID* positions = _positions->host_copy();
          auto rows = std::make_shared<RowIDPosList>();
          for (uint64_t i = begin; i < end; ++i) rows->rows.push_back(positions[i].is_null() ? NULL_ROW_ID : (*_lists[g][positions[i].chunk_id])[positions[i].chunk_offset]);
          of_group[g] = rows;
        }
        const auto first = std::static_pointer_cast<ReferenceSegment>(_input->get_chunk(0)->get_segment(c));
        segments.push_back(std::make_shared<ReferenceSegment>(first->referenced_table(), first->referenced_column_id(), of_group[g]));
      }
    }

   private:
    std::shared_ptr<const Table> _input;
    std::shared_ptr<DeviceBlock> _positions;
    uint64_t _n;
    std::vector<size_t> _group_of_column;
    std::vector<std::shared_ptr<DeviceBlock>> _resolved;
    std::vector<std::vector<const AbstractPosList*>> _lists;
  };

  // write_output_segments (join_output_writing.cpp:95-200): reference inputs are dereferenced through their pos lists.
  static void append_side(Segments& segments, const std::shared_ptr<const Table>& input, std::vector<RowID> positions) {
    if (input->type() == TableType::Data) {
      const auto pos_list = std::make_shared<RowIDPosList>(std::move(positions));
      for (ColumnID c = 0; c < input->column_count(); ++c) segments.push_back(std::make_shared<ReferenceSegment>(input, c, pos_list));
      return;
    }
    std::map<std::vector<const AbstractPosList*>, std::shared_ptr<RowIDPosList>> cache;
    for (ColumnID c = 0; c < input->column_count(); ++c) {
      std::vector<const AbstractPosList*> lists;
      for (ChunkID k = 0; k < input->chunk_count(); ++k) lists.push_back(std::static_pointer_cast<ReferenceSegment>(input->get_chunk(k)->get_segment(c))->pos_list().get());
      auto& resolved = cache[lists];
      if (!resolved) {
        resolved = std::make_shared<RowIDPosList>();
        for (const auto& row : positions) resolved->rows.push_back(row.is_null() ? NULL_ROW_ID : (*lists[row.chunk_id])[row.chunk_offset]);
      }
      const auto first = std::static_pointer_cast<ReferenceSegment>(input->get_chunk(0)->get_segment(c));
      segments.push_back(std::make_shared<ReferenceSegment>(first->referenced_table(), first->referenced_column_id(), resolved));
    }
  }

  JoinMode _mode;
  ColumnIDPair _column_ids;
  std::optional<size_t> _radix_bits;
  std::vector<OperatorJoinPredicate> _secondary_predicates;
};

struct AggregateDefinition {   // WindowFunctionExpression over a PQPColumnExpression (INVALID_COLUMN_ID: COUNT(*))
  ColumnID column_id;
  WindowFunction function;
};

class AggregateHash : public AbstractReadOnlyOperator {   // operators/aggregate_hash.hpp:139-141
 public:
  AggregateHash(std::shared_ptr<const AbstractOperator> in, std::vector<AggregateDefinition> aggregates, std::vector<ColumnID> groupby_column_ids)
      : AbstractReadOnlyOperator(std::move(in)), _aggregates(std::move(aggregates)), _groupby(std::move(groupby_column_ids)) {}
  const std::string& name() const override { static const std::string n = "AggregateHash"; return n; }

 protected:
  std::shared_ptr<const Table> _on_execute() override {
    const auto input = left_input_table();
    std::vector<std::shared_ptr<DeviceColumn>> keep;
    std::vector<const hy_column*> groupby;
    for (const auto id : _groupby) { keep.push_back(device_column(input, id, input->column_data_type(id) == DataType::String ? StringKeys::AggregateKeyNames : StringKeys::None)); groupby.push_back(keep.back()->handle); }
    std::vector<hy_aggregate_spec> specs;
    for (const auto& aggregate : _aggregates) {
      hy_aggregate_spec spec{};
      spec.function = static_cast<uint32_t>(aggregate.function);
      if (aggregate.column_id != INVALID_COLUMN_ID) {
        // SUM / AVG / STDDEV_SAMP of strings are invalid (aggregate_test.cpp:232-262 expects std::logic_error)
        const bool arithmetic = aggregate.function == WindowFunction::Sum || aggregate.function == WindowFunction::Avg || aggregate.function == WindowFunction::StandardDeviationSample;
        Assert(!(arithmetic && input->column_data_type(aggregate.column_id) == DataType::String), "Aggregate function not available for strings.");
        keep.push_back(device_column(input, aggregate.column_id));
        spec.column = keep.back()->handle;
      } else {
        Assert(aggregate.function == WindowFunction::Count, "Only COUNT may have an invalid ColumnID.");   // aggregate_hash.cpp:1002
      }
      specs.push_back(spec);
    }
    std::vector<hy_aggregate_spec> call_specs = specs;
    if (groupby.empty() && std::all_of(specs.begin(), specs.end(), [](const auto& s) { return !s.column; })) {
      // a lone COUNT(*): pass the table's first column as an (ignored) ANY so that the library knows the chunk layout
      keep.push_back(device_column(input, ColumnID{0}));
      hy_aggregate_spec lone_spec{};
      lone_spec.function = HY_AGG_ANY;
      lone_spec.column = keep.back()->handle;
      call_specs.push_back(lone_spec);
    }
    // Room for a group per input row, as the one call may need -- but not TOUCHED: value-initialised vectors of that size were 0.8 GB of page
    // faults (a quarter of a second) in front of an aggregate over 26 M joined rows that produces three groups.
    const uint32_t capacity = static_cast<uint32_t>(input->row_count() + 1);
    const std::unique_ptr<RowID[]> group_rows(new RowID[capacity]);
    std::vector<std::unique_ptr<uint64_t[]>> values;
    std::vector<std::unique_ptr<uint8_t[]>> nulls;
    std::vector<hy_aggregate_column> columns(std::max<size_t>(1, call_specs.size()));
    for (size_t a = 0; a < call_specs.size(); ++a) {
      values.emplace_back(new uint64_t[capacity]);
      nulls.emplace_back(new uint8_t[capacity]);
      columns[a].values = values[a].get();
      columns[a].is_null = nulls[a].get();
    }
    hy_aggregate_result result{};
    result.mem = HY_MEM_HOST;
    result.group_capacity = capacity;
    result.group_row_ids = reinterpret_cast<hy_row_id*>(group_rows.get());
    result.columns = columns.data();
    check_status(hy_aggregate_hash(groupby.data(), static_cast<uint32_t>(groupby.size()), call_specs.data(), static_cast<uint32_t>(call_specs.size()), &result));
    // ---- output (aggregate_hash.cpp:1301-1361): GROUP BY columns reference the input through the representative
    // rows, aggregate columns are ValueSegments; here both are materialised into one Data table of value segments.
    TableColumnDefinitions definitions;
    for (const auto id : _groupby) definitions.push_back(input->column_definitions()[id]);
    static const char* names[] = {"MIN", "MAX", "SUM", "AVG", "COUNT", "COUNT DISTINCT", "STDDEV_SAMP", "ANY"};
    for (size_t a = 0; a < specs.size(); ++a) {
      const auto& aggregate = _aggregates[a];
      const std::string argument = aggregate.column_id == INVALID_COLUMN_ID ? "*" : input->column_name(aggregate.column_id);
      const bool needs_null = aggregate.function != WindowFunction::Count && aggregate.function != WindowFunction::CountDistinct;
      definitions.push_back({std::string(names[static_cast<int>(aggregate.function)]) + "(" + argument + ")", static_cast<DataType>(columns[a].data_type), needs_null});
    }
    auto output = std::make_shared<Table>(definitions, TableType::Data, Chunk::DEFAULT_SIZE);
    for (uint32_t g = 0; g < result.n_groups; ++g) {
      std::vector<AllTypeVariant> row;
      for (const auto id : _groupby) row.push_back((*input->get_chunk(group_rows[g].chunk_id)->get_segment(id))[group_rows[g].chunk_offset]);
      for (size_t a = 0; a < specs.size(); ++a) {
        if (nulls[a][g]) { row.emplace_back(NullValue{}); continue; }
        switch (static_cast<DataType>(columns[a].data_type)) {
          case DataType::Int: row.emplace_back(reinterpret_cast<const int32_t*>(values[a].get())[g]); break;
          case DataType::Long: row.emplace_back(reinterpret_cast<const int64_t*>(values[a].get())[g]); break;
          case DataType::Float: row.emplace_back(reinterpret_cast<const float*>(values[a].get())[g]); break;
          default: row.emplace_back(reinterpret_cast<const double*>(values[a].get())[g]); break;
        }
      }
      output->append(std::move(row));
    }
    output->finalize();
    return output;
  }

 private:
  std::vector<AggregateDefinition> _aggregates;
  std::vector<ColumnID> _groupby;
};

// ---- TableScan(s) -> Projection -> AggregateHash of one data table in ONE pass (hy_scan_project_aggregate) --------------------------------
// What the adapter substitutes for the plan  GetTable -> [Validate ->] TableScan ... -> Projection -> AggregateHash  when every scan is a
// ColumnVsValue / Between / IsNull scan of a numeric column of the stored table and every aggregate a MIN / MAX / SUM / AVG / COUNT of an
// arithmetic expression over its columns (tpch_queries.cpp:60-80, 206-210: Q1, Q6).  Expressions are given in postfix order, like
// hy_expression:  l_extendedprice * (1 - l_discount)  =  column, literal 1, column, Subtraction, Multiplication.
enum class ArithmeticOperator : uint8_t { Addition, Subtraction, Multiplication, Division, Modulo };   // expression/arithmetic_expression.hpp:12
struct ExpressionNode {
  static ExpressionNode column(ColumnID id) { ExpressionNode n; n.kind = HY_EXPR_COLUMN; n.column_id = id; return n; }
  static ExpressionNode literal(AllTypeVariant v) { ExpressionNode n; n.kind = HY_EXPR_LITERAL; n.value = std::move(v); return n; }
  static ExpressionNode arithmetic(ArithmeticOperator op) { ExpressionNode n; n.kind = HY_EXPR_ARITHMETIC; n.op = op; return n; }
  uint32_t kind = HY_EXPR_COLUMN;
  ColumnID column_id = INVALID_COLUMN_ID;
  AllTypeVariant value;
  ArithmeticOperator op = ArithmeticOperator::Addition;
};
struct ScanPredicate {   // column <condition> value [AND value2]
  ColumnID column_id;
  PredicateCondition condition;
  AllTypeVariant value;
  std::optional<AllTypeVariant> value2;
};
struct ExpressionAggregate {   // WindowFunctionExpression over an arithmetic expression (no nodes: COUNT(*))
  WindowFunction function;
  std::vector<ExpressionNode> input;
};

class ScanProjectAggregate : public AbstractReadOnlyOperator {
 public:
  ScanProjectAggregate(std::shared_ptr<const AbstractOperator> in, std::vector<ScanPredicate> predicates, std::vector<ColumnID> groupby_column_ids, std::vector<ExpressionAggregate> aggregates)
      : AbstractReadOnlyOperator(std::move(in)), _predicates(std::move(predicates)), _groupby(std::move(groupby_column_ids)), _aggregates(std::move(aggregates)) {}
  const std::string& name() const override { static const std::string n = "ScanProjectAggregate"; return n; }
  // with a context, Validate is the pass's first filter (sql_pipeline_builder.hpp:55: every SQL-driven plan validates behind GetTable)
  void set_transaction_context(std::shared_ptr<TransactionContext> context) { _context = std::move(context); }

 protected:
  std::shared_ptr<const Table> _on_execute() override {
    const auto input = left_input_table();
    Assert(input->type() == TableType::Data, "ScanProjectAggregate reads a stored table: run the operator chain on reference tables.");
    std::vector<std::shared_ptr<DeviceColumn>> keep;
    std::vector<hy_filter> filters;
    if (_context) {
      keep.push_back(mvcc_column(input));
      hy_filter filter{};
      filter.column = keep.back()->handle;
      filter.predicate.condition = HY_FILTER_VALIDATE;
      filter.predicate.value.value_id = _context->transaction_id();
      filter.predicate.value2.value_id = _context->snapshot_commit_id();
      filter.predicate.column_is_nullable = _context->has_in_flight_delete() ? 0u : 1u;   // can_use_chunk_shortcut (validate.cpp:116-127)
      filters.push_back(filter);
    }
    for (const auto& predicate : _predicates) {
      const auto column_type = input->column_data_type(predicate.column_id);
      Assert(column_type != DataType::String, "ScanProjectAggregate: string predicates run as the operator chain.");
      keep.push_back(device_column(input, predicate.column_id));
      hy_filter filter{};
      filter.column = keep.back()->handle;
      filter.predicate.condition = static_cast<uint32_t>(predicate.condition);
      filter.predicate.column_is_nullable = input->column_is_nullable(predicate.column_id);
      if (predicate.condition != PredicateCondition::IsNull && predicate.condition != PredicateCondition::IsNotNull) {
        const hy_value first = to_hy_value(predicate.value), second = predicate.value2 ? to_hy_value(*predicate.value2) : hy_value{};
        hy_predicate cast{};
        const auto status = hy_predicate_cast(static_cast<uint32_t>(predicate.condition), static_cast<uint32_t>(column_type), static_cast<uint32_t>(data_type_from_all_type_variant(predicate.value)), &first,
                                              predicate.value2 ? static_cast<uint32_t>(data_type_from_all_type_variant(*predicate.value2)) : HY_TYPE_NULL, predicate.value2 ? &second : nullptr, &cast);
        Assert(status != HY_ERR_UNSUPPORTED, "ScanProjectAggregate: the literal has no lossless predicate cast -- run the operator chain.");
        check_status(status);
        filter.predicate.condition = cast.condition, filter.predicate.value_type = cast.value_type, filter.predicate.value = cast.value, filter.predicate.value2 = cast.value2;
      }
      filters.push_back(filter);
    }
    std::vector<const hy_column*> groupby;
    for (const auto id : _groupby) {
      keep.push_back(device_column(input, id, input->column_data_type(id) == DataType::String ? StringKeys::AggregateKeyNames : StringKeys::None));
      groupby.push_back(keep.back()->handle);
    }
    std::vector<hy_expression> expressions(_aggregates.size());
    std::vector<hy_fused_aggregate> specs(_aggregates.size());
    for (size_t a = 0; a < _aggregates.size(); ++a) {
      specs[a].function = static_cast<uint32_t>(_aggregates[a].function);
      const auto& nodes = _aggregates[a].input;
      if (nodes.empty()) { Assert(_aggregates[a].function == WindowFunction::Count, "Only COUNT may omit its argument."); continue; }
      Assert(nodes.size() <= HY_MAX_EXPRESSION_NODES, "ScanProjectAggregate: expression too long -- run the operator chain.");
      expressions[a].n_nodes = static_cast<uint32_t>(nodes.size());
      for (size_t k = 0; k < nodes.size(); ++k) {
        hy_expression_node& out = expressions[a].nodes[k];
        out.kind = nodes[k].kind;
        if (nodes[k].kind == HY_EXPR_COLUMN) {
          keep.push_back(device_column(input, nodes[k].column_id));
          out.column = keep.back()->handle;
        } else if (nodes[k].kind == HY_EXPR_LITERAL) {
          out.literal_type = static_cast<uint32_t>(data_type_from_all_type_variant(nodes[k].value));
          if (!variant_is_null(nodes[k].value)) out.literal = to_hy_value(nodes[k].value);
        } else {
          out.op = static_cast<uint32_t>(nodes[k].op);
        }
      }
      specs[a].input = &expressions[a];
    }
    const uint32_t capacity = static_cast<uint32_t>(input->row_count() + 1);
    std::vector<RowID> group_rows(capacity);
    std::vector<std::vector<uint64_t>> values(specs.size(), std::vector<uint64_t>(capacity));
    std::vector<std::vector<uint8_t>> nulls(specs.size(), std::vector<uint8_t>(capacity));
    std::vector<hy_aggregate_column> columns(std::max<size_t>(1, specs.size()));
    for (size_t a = 0; a < specs.size(); ++a) { columns[a].values = values[a].data(); columns[a].is_null = nulls[a].data(); }
    hy_aggregate_result result{};
    result.mem = HY_MEM_HOST;
    result.group_capacity = capacity;
    result.group_row_ids = reinterpret_cast<hy_row_id*>(group_rows.data());
    result.columns = columns.data();
    check_status(hy_scan_project_aggregate(filters.data(), static_cast<uint32_t>(filters.size()), groupby.data(), static_cast<uint32_t>(groupby.size()), specs.data(),
                                           static_cast<uint32_t>(specs.size()), &result));
    // output like AggregateHash's: GROUP BY values through the representative rows (rows of the stored table), then the aggregates
    TableColumnDefinitions definitions;
    for (const auto id : _groupby) definitions.push_back(input->column_definitions()[id]);
    static const char* names[] = {"MIN", "MAX", "SUM", "AVG", "COUNT", "COUNT DISTINCT", "STDDEV_SAMP", "ANY"};
    for (size_t a = 0; a < specs.size(); ++a) {
      const bool needs_null = _aggregates[a].function != WindowFunction::Count;
      definitions.push_back({std::string(names[static_cast<int>(_aggregates[a].function)]) + "(" + (_aggregates[a].input.empty() ? "*" : "expression " + std::to_string(a)) + ")",
                             static_cast<DataType>(columns[a].data_type), needs_null});
    }
    auto output = std::make_shared<Table>(definitions, TableType::Data, Chunk::DEFAULT_SIZE);
    for (uint32_t g = 0; g < result.n_groups; ++g) {
      std::vector<AllTypeVariant> row;
      for (const auto id : _groupby) row.push_back((*input->get_chunk(group_rows[g].chunk_id)->get_segment(id))[group_rows[g].chunk_offset]);
      for (size_t a = 0; a < specs.size(); ++a) {
        if (nulls[a][g]) { row.emplace_back(NullValue{}); continue; }
        switch (static_cast<DataType>(columns[a].data_type)) {
          case DataType::Int: row.emplace_back(reinterpret_cast<const int32_t*>(values[a].data())[g]); break;
          case DataType::Long: row.emplace_back(reinterpret_cast<const int64_t*>(values[a].data())[g]); break;
          case DataType::Float: row.emplace_back(reinterpret_cast<const float*>(values[a].data())[g]); break;
          default: row.emplace_back(reinterpret_cast<const double*>(values[a].data())[g]); break;
        }
      }
      output->append(std::move(row));
    }
    output->finalize();
    return output;
  }

 private:
  std::vector<ScanPredicate> _predicates;
  std::vector<ColumnID> _groupby;
  std::vector<ExpressionAggregate> _aggregates;
  std::shared_ptr<TransactionContext> _context;
};

// ---- a star join as one operator (hy_star_join_aggregate, csrc/plan.hip) ---------------------------------------------------------------
// fact JOIN dimension_1 ... JOIN dimension_k (Inner, one key each, in this order) -> GROUP BY columns of any of the tables -> aggregates
// over fact / dimension columns or over `left <op> right`: the subtree TableScan(s) -> JoinHash(es) -> Projection -> AggregateHash that
// Hyrise's optimizer produces for SSB, handed to the library as ONE call.  Inputs are stored tables; the output is AggregateHash's: the
// GROUP BY columns, then one column per aggregate.
struct StarDimension {
  std::shared_ptr<const AbstractOperator> table;
  ColumnID key;
  std::optional<ScanPredicate> filter;   // on a column of the dimension; none: the dimension is joined whole
  ColumnID fact_key;
};
struct StarColumn { size_t table; ColumnID column; };   // table 0 = the fact table, d + 1 = dimension d
struct StarAggregate {
  WindowFunction function;
  std::optional<StarColumn> left;                    // none: COUNT(*)
  std::optional<ArithmeticOperator> op;              // with `right`: the aggregate of  left <op> right
  std::optional<StarColumn> right;
};

class StarJoinAggregate : public AbstractReadOnlyOperator {
 public:
  StarJoinAggregate(std::shared_ptr<const AbstractOperator> fact, std::vector<StarDimension> dimensions, std::vector<StarColumn> groupby, std::vector<StarAggregate> aggregates)
      : AbstractReadOnlyOperator(std::move(fact)), _dimensions(std::move(dimensions)), _groupby(std::move(groupby)), _aggregates(std::move(aggregates)) {}
  const std::string& name() const override { static const std::string n = "StarJoinAggregate"; return n; }
  uint64_t joined_rows = 0;

 protected:
  std::shared_ptr<const Table> _on_execute() override {
    const auto fact = left_input_table();
    std::vector<std::shared_ptr<const Table>> tables{fact};
    for (const auto& dimension : _dimensions) tables.push_back(dimension.table->get_output());   // (executed by the caller, like every input)
    for (const auto& table : tables) Assert(table->type() == TableType::Data, "StarJoinAggregate reads stored tables.");
    std::vector<std::shared_ptr<DeviceColumn>> keep;
    auto column_of = [&](const StarColumn& c) -> const hy_column* {
      Assert(c.table < tables.size(), "StarJoinAggregate: no such table.");
      Assert(tables[c.table]->column_data_type(c.column) != DataType::String, "StarJoinAggregate: string columns run as the operator chain.");
      keep.push_back(device_column(tables[c.table], c.column));
      return keep.back()->handle;
    };
    std::vector<hy_star_dimension> dims(_dimensions.size());
    for (size_t d = 0; d < _dimensions.size(); ++d) {
      std::memset(&dims[d], 0, sizeof(dims[d]));
      dims[d].key = column_of({d + 1, _dimensions[d].key});
      dims[d].fact_key = column_of({0, _dimensions[d].fact_key});
      if (!_dimensions[d].filter) continue;
      const ScanPredicate& predicate = *_dimensions[d].filter;
      const auto column_type = tables[d + 1]->column_data_type(predicate.column_id);
      dims[d].filter_column = column_of({d + 1, predicate.column_id});
      dims[d].predicate.condition = static_cast<uint32_t>(predicate.condition);
      dims[d].predicate.column_is_nullable = tables[d + 1]->column_is_nullable(predicate.column_id);
      if (predicate.condition != PredicateCondition::IsNull && predicate.condition != PredicateCondition::IsNotNull) {
        const hy_value first = to_hy_value(predicate.value), second = predicate.value2 ? to_hy_value(*predicate.value2) : hy_value{};
        hy_predicate cast{};
        const auto status = hy_predicate_cast(static_cast<uint32_t>(predicate.condition), static_cast<uint32_t>(column_type), static_cast<uint32_t>(data_type_from_all_type_variant(predicate.value)), &first,
                                              predicate.value2 ? static_cast<uint32_t>(data_type_from_all_type_variant(*predicate.value2)) : HY_TYPE_NULL, predicate.value2 ? &second : nullptr, &cast);
        Assert(status != HY_ERR_UNSUPPORTED, "StarJoinAggregate: the literal has no lossless predicate cast -- run the operator chain.");
        check_status(status);
        dims[d].predicate.condition = cast.condition, dims[d].predicate.value_type = cast.value_type, dims[d].predicate.value = cast.value, dims[d].predicate.value2 = cast.value2;
      }
    }
    std::vector<hy_star_column> groupby(_groupby.size());
    for (size_t g = 0; g < _groupby.size(); ++g) groupby[g] = hy_star_column{static_cast<uint32_t>(_groupby[g].table), 0, column_of(_groupby[g])};
    // the caller's aggregates, then MIN of every GROUP BY column: the groups' representative rows are rows of an intermediate table
    std::vector<hy_star_aggregate> specs(_aggregates.size() + _groupby.size());
    for (size_t a = 0; a < _aggregates.size(); ++a) {
      std::memset(&specs[a], 0, sizeof(specs[a]));
      specs[a].function = static_cast<uint32_t>(_aggregates[a].function);
      specs[a].op = HY_STAR_NO_OP;
      if (!_aggregates[a].left) { Assert(_aggregates[a].function == WindowFunction::Count, "Only COUNT may omit its argument."); continue; }
      specs[a].left = hy_star_column{static_cast<uint32_t>(_aggregates[a].left->table), 0, column_of(*_aggregates[a].left)};
      if (_aggregates[a].op) {
        Assert(_aggregates[a].right.has_value(), "StarJoinAggregate: an expression needs two columns.");
        specs[a].op = static_cast<uint32_t>(*_aggregates[a].op);
        specs[a].right = hy_star_column{static_cast<uint32_t>(_aggregates[a].right->table), 0, column_of(*_aggregates[a].right)};
      }
    }
    for (size_t g = 0; g < _groupby.size(); ++g) {
      std::memset(&specs[_aggregates.size() + g], 0, sizeof(hy_star_aggregate));
      specs[_aggregates.size() + g].function = HY_AGG_MIN;
      specs[_aggregates.size() + g].op = HY_STAR_NO_OP;
      specs[_aggregates.size() + g].left = groupby[g];
    }
    Assert(specs.size() <= HY_MAX_STAR_AGGREGATES, "StarJoinAggregate: too many aggregates and GROUP BY columns for one call.");
    const uint32_t capacity = static_cast<uint32_t>(std::min<uint64_t>(fact->row_count() + 1, 1u << 20));
    std::vector<RowID> group_rows(capacity);
    std::vector<std::vector<uint64_t>> values(specs.size(), std::vector<uint64_t>(capacity));
    std::vector<std::vector<uint8_t>> nulls(specs.size(), std::vector<uint8_t>(capacity));
    std::vector<hy_aggregate_column> columns(std::max<size_t>(1, specs.size()));
    for (size_t a = 0; a < specs.size(); ++a) { columns[a].values = values[a].data(); columns[a].is_null = nulls[a].data(); }
    hy_aggregate_result result{};
    result.mem = HY_MEM_HOST;
    result.group_capacity = capacity;
    result.group_row_ids = reinterpret_cast<hy_row_id*>(group_rows.data());
    result.columns = columns.data();
    check_status(hy_star_join_aggregate(dims.data(), static_cast<uint32_t>(dims.size()), groupby.data(), static_cast<uint32_t>(groupby.size()), specs.data(), static_cast<uint32_t>(specs.size()),
                                        &result, &joined_rows));
    TableColumnDefinitions definitions;
    for (const auto& g : _groupby) definitions.push_back(tables[g.table]->column_definitions()[g.column]);
    static const char* names[] = {"MIN", "MAX", "SUM", "AVG", "COUNT", "COUNT DISTINCT", "STDDEV_SAMP", "ANY"};
    for (size_t a = 0; a < _aggregates.size(); ++a)
      definitions.push_back({std::string(names[static_cast<int>(_aggregates[a].function)]) + "(" + (_aggregates[a].left ? "expression " + std::to_string(a) : "*") + ")", static_cast<DataType>(columns[a].data_type),
                             _aggregates[a].function != WindowFunction::Count});
    auto cell = [&](size_t a, uint32_t g) -> AllTypeVariant {
      if (nulls[a][g]) return NullValue{};
      switch (static_cast<DataType>(columns[a].data_type)) {
        case DataType::Int: return reinterpret_cast<const int32_t*>(values[a].data())[g];
        case DataType::Long: return reinterpret_cast<const int64_t*>(values[a].data())[g];
        case DataType::Float: return reinterpret_cast<const float*>(values[a].data())[g];
        default: return reinterpret_cast<const double*>(values[a].data())[g];
      }
    };
    auto output = std::make_shared<Table>(definitions, TableType::Data, Chunk::DEFAULT_SIZE);
    for (uint32_t g = 0; g < result.n_groups; ++g) {
      std::vector<AllTypeVariant> row;
      for (size_t k = 0; k < _groupby.size(); ++k) row.push_back(cell(_aggregates.size() + k, g));
      for (size_t a = 0; a < _aggregates.size(); ++a) row.push_back(cell(a, g));
      output->append(std::move(row));
    }
    output->finalize();
    return output;
  }

 private:
  std::vector<StarDimension> _dimensions;
  std::vector<StarColumn> _groupby;
  std::vector<StarAggregate> _aggregates;
};

}  // namespace hyrise_amd
