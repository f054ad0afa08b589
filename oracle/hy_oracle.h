/*
 * hy_oracle.h -- CPU restatement of the Hyrise hot path (TableScan / JoinHash / AggregateHash).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so, and only as the checker / reported CPU baseline.
 *
 * The reference (C++23 + Boost + oneTBB + 18 absent submodules) cannot be compiled in this environment
 * (SURVEY.md section 0), so this is a restatement that follows the reference function by function; every function
 * cites the file:line it follows.  It is pinned against the reference's own fixtures and known-answer tests
 * (tests/golden/, see tests/test_oracle_*.py); what those tests do NOT pin (PosList order, join pair order,
 * group order) is defined by the cited reference code only -- see DESIGN.md section 5.
 *
 * Descriptors are the product ABI's plain structs (include/hyrise_amd.h) with HOST pointers; `hy_segment.ref`
 * points to an `hyo_column` here.
 */
#ifndef HY_ORACLE_H_
#define HY_ORACLE_H_

#include "../include/hyrise_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hyo_column {
  const hy_segment* segments;
  uint32_t n_chunks;
} hyo_column;

/* ---- encoders (dictionary_encoder.hpp:33-103, frame_of_reference_encoder.hpp:25-122,
 *      fixed_width_integer_compressor.cpp:33-57) ---------------------------------------------------------------- */
/* values: T[n] (NULL slots ignored), nulls: byte-per-row or NULL.  dict_out capacity n.  Returns dictionary size;
 * *width_out = 1|2|4 (attribute vector element width), av_out capacity n*4 bytes. */
uint32_t hyo_encode_dictionary(uint32_t data_type, const void* values, const uint8_t* nulls, uint32_t n,
                               void* dict_out, void* av_out, uint32_t* width_out);
/* int32 only.  minima_out capacity ceil(n/2048); offsets_out capacity n*4 bytes.  Returns width (1|2|4);
 * *has_nulls_out = whether the segment keeps a null vector. */
uint32_t hyo_encode_frame_of_reference(const int32_t* values, const uint8_t* nulls, uint32_t n, int32_t* minima_out,
                                       void* offsets_out, uint32_t* has_nulls_out);
/* byte-per-row null flags -> libstdc++ vector<bool> words. */
void hyo_pack_nulls(const uint8_t* nulls, uint32_t n, uint64_t* words_out);

/* ---- TableScan ------------------------------------------------------------------------------------------------ */
/* AbstractDereferencedColumnTableScanImpl::scan_chunk for ColumnVsValue / ColumnBetween / ColumnIsNull.
 * matches capacity = segment size.  Returns match count, -1 on unsupported input.  state_out (may be NULL) receives
 * HY_CHUNK_ALL_MATCH / HY_CHUNK_NONE_MATCH when the reference takes an early-out, else HY_CHUNK_SCANNED. */
int64_t hyo_scan_chunk(const hyo_column* column, uint32_t chunk_id, const hy_predicate* predicate, hy_row_id* matches,
                       uint8_t* state_out);
/* ColumnVsColumnTableScanImpl::scan_chunk. */
int64_t hyo_scan_chunk_columns(const hyo_column* left, const hyo_column* right, uint32_t chunk_id, uint32_t condition,
                               hy_row_id* matches);
/* Whole-column driver with the ABI's result layout (host memory), single thread or `threads` pthreads over chunks
 * (JobTask fan-out, table_scan.cpp:223-229). */
int32_t hyo_table_scan(const hyo_column* column, const hy_predicate* predicate, hy_scan_result* result, int threads);
/* The literal handling in front of a TableScan (predicate_cast.c): lossless_predicate_cast.{hpp,cpp}, table_scan.cpp:336-448. */
int hyo_next_float_towards(double value, double towards, float* out);
int hyo_lossless_predicate_cast(uint32_t condition, uint32_t source_type, const hy_value* value, uint32_t target_type, uint32_t* out_condition,
                                hy_value* out_value);
int hyo_predicate_for_column(uint32_t condition, uint32_t column_type, uint32_t value_type, const hy_value* value, uint32_t value2_type,
                             const hy_value* value2, hy_predicate* out);
/* Projection arithmetic (projection.c). */
uint32_t hyo_expression_common_type(uint32_t lhs, uint32_t rhs);
int hyo_arithmetic_cell(uint32_t op, uint32_t a_type, const void* a, int a_null, uint32_t b_type, const void* b, int b_null, void* result);
uint32_t hyo_arithmetic(uint32_t op, uint32_t a_type, const void* a, const uint8_t* a_nulls, uint32_t a_stride, uint32_t b_type, const void* b,
                        const uint8_t* b_nulls, uint32_t b_stride, uint64_t n, void* result, uint8_t* result_nulls);
/* Validate (validate.c): visible positions per input chunk, same result layout as hyo_table_scan. */
int32_t hyo_validate(const hyo_column* column, uint32_t our_tid, uint32_t snapshot_commit_id, uint32_t can_use_chunk_shortcut,
                     hy_scan_result* result);
int32_t hyo_table_scan_columns(const hyo_column* left, const hyo_column* right, uint32_t condition,
                               hy_scan_result* result, int threads);

/* ---- JoinHash ------------------------------------------------------------------------------------------------- */
int32_t hyo_join_hash(const hyo_column* left, const hyo_column* right, uint32_t mode, hy_join_result* result,
                      int threads);
/* With secondary predicates (the hy_join_predicate's columns are hyo_column pointers here). */
int32_t hyo_join_hash_predicates(const hyo_column* left, const hyo_column* right, uint32_t mode, const hy_join_predicate* secondary,
                                 uint32_t n_secondary, hy_join_result* result, int threads);
uint32_t hyo_calculate_radix_bits(uint64_t build_rows, uint64_t probe_rows);
/* write_output_chunks' chunking of the per-partition PosLists incl. the MIN_SIZE / MAX_SIZE merge (join_output_writing.cpp:245-296). */
uint32_t hyo_write_output_chunks(const uint64_t* slice_offsets, uint32_t n_slices, int allow_partition_merge, uint64_t* chunk_offsets);
/* Step-level entry points pinned by join_hash_steps_test.cpp. */
/* materialize_input: writes (row_id,value) elements of one column in chunk order; returns element count.
 * bloom_in may be NULL (all-true); bloom_out 2^20 bits = 16384 u64 words (zeroed by the caller).
 * histograms_out [n_chunks << radix_bits] */
/* std::hash<HashedType>{}(key) as libstdc++ computes it; key: an integer, or the bit pattern of a float (zero-extended) /
 * double with -0.0 given as +0.0.  hashed_type: HY_TYPE_*. */
uint32_t hyo_join_hashed_type(uint32_t left_type, uint32_t right_type);   /* JoinHashTraits<L, R>::HashType, join_hash_traits.hpp:15-40 (numeric types) */
uint64_t hyo_std_hash(int64_t key, uint32_t hashed_type);
uint64_t hyo_std_hash_bytes(const void* data, uint64_t length);   /* std::hash<std::string / pmr_string> over the bytes */
uint64_t hyo_join_materialize(const hyo_column* column, int keep_nulls, uint32_t radix_bits, const uint64_t* bloom_in,
                              uint64_t* bloom_out, hy_row_id* row_ids_out, int64_t* values_out, uint8_t* nulls_out,
                              uint64_t* chunk_element_counts_out, uint64_t* histograms_out);

/* ---- AggregateHash -------------------------------------------------------------------------------------------- */
int32_t hyo_aggregate_hash(const hyo_column* const* groupby_columns, uint32_t n_groupby,
                           const uint32_t* functions, const hyo_column* const* aggregate_columns, uint32_t n_aggregates,
                           hy_aggregate_result* result);

#ifdef __cplusplus
}
#endif
#endif
