/*
 * scan.c -- restatement of TableScan's per-chunk scan implementations.  TEST INFRASTRUCTURE ONLY (see hy_oracle.h).
 *
 * Follows, function by function:
 *   AbstractTableScanImpl::_scan_with_iterators            operators/table_scan/abstract_table_scan_impl.hpp:44-84
 *   AbstractDereferencedColumnTableScanImpl::scan_chunk,
 *     _scan_reference_segment                              abstract_dereferenced_column_table_scan_impl.cpp:19-107
 *   split_pos_list_by_chunk_id                             storage/split_pos_list_by_chunk_id.cpp:13-59
 *   ColumnVsValueTableScanImpl                             column_vs_value_table_scan_impl.cpp:43-272 (+ .hpp:57-81)
 *   ColumnBetweenTableScanImpl                             column_between_table_scan_impl.cpp:42-195
 *   with_between_comparator                                type_comparison.hpp:114-159
 *   ColumnIsNullTableScanImpl                              column_is_null_table_scan_impl.cpp:37-258
 *   ColumnVsColumnTableScanImpl                            column_vs_column_table_scan_impl.cpp:36-187
 * Sorted-segment binary search (SortedSegmentSearch) is not restated: TPC-H data is unclustered by default and no
 * chunk carries a sort flag; the descriptors carry no sort information.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "hy_oracle.h"

static inline uint32_t load_compressed(const void* data, uint32_t width, uint32_t i) {
  if (width == 1) return ((const uint8_t*)data)[i];
  if (width == 2) return ((const uint16_t*)data)[i];
  return ((const uint32_t*)data)[i];
}

static inline int bitmap_get(const uint64_t* words, uint32_t i) {
  return (int)((words[i / 64] >> (i % 64)) & 1u);
}

/* NULL test of a data segment at offset i. */
static inline int seg_is_null(const hy_segment* s, uint32_t i) {
  if (s->encoding == HY_ENC_DICTIONARY) return load_compressed(s->data, s->width, i) == s->aux_size;
  return s->nulls ? bitmap_get(s->nulls, i) : 0;
}

/* Decoded value as double-width carriers (exact for every supported type: int32/int64 via i64, float/double via
 * f64 -- float -> double is exact). */
typedef struct {
  int64_t i;
  double f;
} decoded_t;

static inline decoded_t seg_value(const hy_segment* s, uint32_t i) {
  decoded_t v = {0, 0.0};
  switch (s->encoding) {
    case HY_ENC_UNENCODED:
      switch (s->data_type) {
        case HY_TYPE_INT: v.i = ((const int32_t*)s->data)[i]; break;
        case HY_TYPE_LONG: v.i = ((const int64_t*)s->data)[i]; break;
        case HY_TYPE_FLOAT: v.f = ((const float*)s->data)[i]; break;
        case HY_TYPE_DOUBLE: v.f = ((const double*)s->data)[i]; break;
        default: break;
      }
      break;
    case HY_ENC_DICTIONARY: {
      const uint32_t vid = load_compressed(s->data, s->width, i);
      if (vid >= s->aux_size || !s->aux) break;
      switch (s->data_type) {
        case HY_TYPE_INT: v.i = ((const int32_t*)s->aux)[vid]; break;
        case HY_TYPE_LONG: v.i = ((const int64_t*)s->aux)[vid]; break;
        case HY_TYPE_FLOAT: v.f = ((const float*)s->aux)[vid]; break;
        case HY_TYPE_DOUBLE: v.f = ((const double*)s->aux)[vid]; break;
        default: break;
      }
      break;
    }
    case HY_ENC_FRAME_OF_REFERENCE: {
      /* frame_of_reference_segment_iterable.hpp:131-139: static_cast<T>(offset) + minimum */
      const uint32_t off = load_compressed(s->data, s->width, i);
      v.i = (int32_t)((uint32_t)off + (uint32_t)((const int32_t*)s->aux)[i / HY_FOR_BLOCK_SIZE]);
      break;
    }
    default: break;
  }
  return v;
}

static inline int type_is_float(uint32_t t) { return t == HY_TYPE_FLOAT || t == HY_TYPE_DOUBLE; }

static inline int cmp_i64(uint32_t cond, int64_t a, int64_t b) {
  switch (cond) {
    case HY_PRED_EQUALS: return a == b;
    case HY_PRED_NOT_EQUALS: return a != b;
    case HY_PRED_LESS_THAN: return a < b;
    case HY_PRED_LESS_THAN_EQUALS: return a <= b;
    case HY_PRED_GREATER_THAN: return a > b;
    case HY_PRED_GREATER_THAN_EQUALS: return a >= b;
    default: return 0;
  }
}
static inline int cmp_f64(uint32_t cond, double a, double b) {
  switch (cond) {
    case HY_PRED_EQUALS: return a == b;
    case HY_PRED_NOT_EQUALS: return a != b;
    case HY_PRED_LESS_THAN: return a < b;
    case HY_PRED_LESS_THAN_EQUALS: return a <= b;
    case HY_PRED_GREATER_THAN: return a > b;
    case HY_PRED_GREATER_THAN_EQUALS: return a >= b;
    default: return 0;
  }
}
static inline int cmp_f32(uint32_t cond, float a, float b) {
  switch (cond) {
    case HY_PRED_EQUALS: return a == b;
    case HY_PRED_NOT_EQUALS: return a != b;
    case HY_PRED_LESS_THAN: return a < b;
    case HY_PRED_LESS_THAN_EQUALS: return a <= b;
    case HY_PRED_GREATER_THAN: return a > b;
    case HY_PRED_GREATER_THAN_EQUALS: return a >= b;
    default: return 0;
  }
}

static inline int is_between(uint32_t c) { return c >= HY_PRED_BETWEEN_INCLUSIVE && c <= HY_PRED_BETWEEN_EXCLUSIVE; }
static inline int lower_inclusive(uint32_t c) {
  return c == HY_PRED_BETWEEN_INCLUSIVE || c == HY_PRED_BETWEEN_UPPER_EXCLUSIVE;
}
static inline int upper_inclusive(uint32_t c) {
  return c == HY_PRED_BETWEEN_INCLUSIVE || c == HY_PRED_BETWEEN_LOWER_EXCLUSIVE;
}

static inline decoded_t literal_of(uint32_t value_type, const hy_value* v) {
  decoded_t d = {0, 0.0};
  switch (value_type) {
    case HY_TYPE_INT: d.i = v->i32; break;
    case HY_TYPE_LONG: d.i = v->i64; break;
    case HY_TYPE_FLOAT: d.f = v->f32; break;
    case HY_TYPE_DOUBLE: d.f = v->f64; break;
    default: break;
  }
  return d;
}

/* DictionarySegment<T>::lower_bound / upper_bound (dictionary_segment.cpp:94-119): INVALID_VALUE_ID at the end. */
static uint32_t dict_bound(const hy_segment* s, decoded_t lit, int upper) {
  uint32_t lo = 0, hi = s->aux_size;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    int go_right;
    switch (s->data_type) {
      case HY_TYPE_INT: { const int64_t e = ((const int32_t*)s->aux)[mid]; go_right = upper ? !(lit.i < e) : (e < lit.i); break; }
      case HY_TYPE_LONG: { const int64_t e = ((const int64_t*)s->aux)[mid]; go_right = upper ? !(lit.i < e) : (e < lit.i); break; }
      case HY_TYPE_FLOAT: { const float e = ((const float*)s->aux)[mid]; const float l = (float)lit.f; go_right = upper ? !(l < e) : (e < l); break; }
      default: { const double e = ((const double*)s->aux)[mid]; go_right = upper ? !(lit.f < e) : (e < lit.f); break; }
    }
    if (go_right) lo = mid + 1; else hi = mid;
  }
  return lo == s->aux_size ? HY_INVALID_VALUE_ID : lo;
}

static int dict_value_equals(const hy_segment* s, uint32_t vid, decoded_t lit) {
  switch (s->data_type) {
    case HY_TYPE_INT: return ((const int32_t*)s->aux)[vid] == lit.i;
    case HY_TYPE_LONG: return ((const int64_t*)s->aux)[vid] == lit.i;
    case HY_TYPE_FLOAT: return ((const float*)s->aux)[vid] == (float)lit.f;
    default: return ((const double*)s->aux)[vid] == lit.f;
  }
}

/* Positions: sequential, or a single-chunk position filter (PointAccessibleSegmentIterable, segment_iterables.hpp).
 * The emitted chunk_offset is the index into the filter (abstract_segment_iterators.hpp: _chunk_offsets()). */
typedef struct {
  const hy_row_id* filter; /* NULL = sequential */
  uint32_t count;
} positions_t;
static inline uint32_t pos_offset(const positions_t* p, uint32_t i) { return p->filter ? p->filter[i].chunk_offset : i; }

static inline void emit(hy_row_id* matches, int64_t* n, uint32_t chunk_id, uint32_t chunk_offset) {
  matches[*n].chunk_id = chunk_id;
  matches[*n].chunk_offset = chunk_offset;
  ++*n;
}

/* ---- ColumnVsValue ------------------------------------------------------------------------------------------- */
static int scan_vs_value_dictionary(const hy_segment* s, uint32_t data_chunk_id, uint32_t out_chunk_id,
                                    const hy_predicate* p, const positions_t* pos, hy_row_id* matches, int64_t* n,
                                    uint8_t* state) {
  const uint32_t cond = p->condition;
  const uint32_t d = s->aux_size;
  uint32_t search;
  int found;
  const int use_upper = (cond == HY_PRED_LESS_THAN_EQUALS || cond == HY_PRED_GREATER_THAN);
  if (s->aux && s->data_type != HY_TYPE_STRING) {
    /* _get_search_value_id (column_vs_value_table_scan_impl.cpp:211-226) */
    const decoded_t lit = literal_of(p->value_type, &p->value);
    search = dict_bound(s, lit, use_upper);
    found = search != HY_INVALID_VALUE_ID && dict_value_equals(s, search, lit);
  } else {
    if (!p->per_chunk_lower || !p->per_chunk_upper) return -1;
    search = use_upper ? p->per_chunk_upper[data_chunk_id] : p->per_chunk_lower[data_chunk_id];
    found = p->per_chunk_found ? p->per_chunk_found[data_chunk_id] : 0;
  }
  int all, none;
  switch (cond) { /* _value_matches_all / _value_matches_none (:228-272) */
    case HY_PRED_EQUALS: all = found && d == 1; none = !found; break;
    case HY_PRED_NOT_EQUALS: all = !found; none = found && d == 1; break;
    case HY_PRED_LESS_THAN:
    case HY_PRED_LESS_THAN_EQUALS: all = search == HY_INVALID_VALUE_ID; none = search == 0; break;
    case HY_PRED_GREATER_THAN:
    case HY_PRED_GREATER_THAN_EQUALS: all = search == 0; none = search == HY_INVALID_VALUE_ID; break;
    default: return -1;
  }
  if (all) { /* :124-155 */
    if (p->column_is_nullable) {
      for (uint32_t i = 0; i < pos->count; ++i) {
        if (load_compressed(s->data, s->width, pos_offset(pos, i)) != d) emit(matches, n, out_chunk_id, i);
      }
    } else {
      for (uint32_t i = 0; i < pos->count; ++i) emit(matches, n, out_chunk_id, i);
      if (state) *state = HY_CHUNK_ALL_MATCH;
    }
    return 0;
  }
  if (none) { /* :157-160 */
    if (state) *state = HY_CHUNK_NONE_MATCH;
    return 0;
  }
  for (uint32_t i = 0; i < pos->count; ++i) { /* :162-179, operators per .hpp:57-81 */
    const uint32_t vid = load_compressed(s->data, s->width, pos_offset(pos, i));
    int match;
    switch (cond) {
      case HY_PRED_EQUALS: match = vid == search; break;
      case HY_PRED_LESS_THAN:
      case HY_PRED_LESS_THAN_EQUALS: match = vid < search; break;
      case HY_PRED_NOT_EQUALS: match = vid != d && vid != search; break;
      default: match = vid != d && vid >= search; break;
    }
    if (match) emit(matches, n, out_chunk_id, i);
  }
  return 0;
}

/* ---- ColumnLike on dictionary segments (column_like_table_scan_impl.cpp:69-121) -------------------------------- */
static int scan_like_dictionary(const hy_segment* s, uint32_t data_chunk_id, uint32_t out_chunk_id, const hy_predicate* p,
                                const positions_t* pos, hy_row_id* matches, int64_t* n, uint8_t* state) {
  if (!p->match_words || !p->match_word_offsets) return -1;
  const uint64_t* dictionary_matches = p->match_words + p->match_word_offsets[data_chunk_id]; /* _find_matches_in_dictionary */
  const uint32_t d = s->aux_size;
  uint32_t match_count = 0;
  for (uint32_t v = 0; v < d; ++v) match_count += (uint32_t)bitmap_get(dictionary_matches, v);
  if (match_count == 0) { /* "LIKE matches no rows" (:108-112) */
    if (state) *state = HY_CHUNK_NONE_MATCH;
    return 0;
  }
  /* match_count == d: "LIKE matches all rows, but we still need to check for NULL" (:96-106) -- the same loop */
  for (uint32_t i = 0; i < pos->count; ++i) {
    const uint32_t vid = load_compressed(s->data, s->width, pos_offset(pos, i));
    if (vid < d && bitmap_get(dictionary_matches, vid)) emit(matches, n, out_chunk_id, i);
  }
  return 0;
}

static int scan_vs_value_generic(const hy_segment* s, uint32_t out_chunk_id, const hy_predicate* p,
                                 const positions_t* pos, hy_row_id* matches, int64_t* n) {
  /* _scan_generic_segment (:64-87): _scan_with_iterators<true>(cmp(value, typed_value)) */
  if (s->data_type == HY_TYPE_STRING) return -1;
  const decoded_t lit = literal_of(p->value_type, &p->value);
  for (uint32_t i = 0; i < pos->count; ++i) {
    const uint32_t off = pos_offset(pos, i);
    if (seg_is_null(s, off)) continue;
    const decoded_t v = seg_value(s, off);
    int match;
    if (s->data_type == HY_TYPE_FLOAT) match = cmp_f32(p->condition, (float)v.f, (float)lit.f);
    else if (s->data_type == HY_TYPE_DOUBLE) match = cmp_f64(p->condition, v.f, lit.f);
    else match = cmp_i64(p->condition, v.i, lit.i);
    if (match) emit(matches, n, out_chunk_id, i);
  }
  return 0;
}

/* ---- ColumnBetween -------------------------------------------------------------------------------------------- */
static int scan_between_dictionary(const hy_segment* s, uint32_t data_chunk_id, uint32_t out_chunk_id,
                                   const hy_predicate* p, const positions_t* pos, hy_row_id* matches, int64_t* n,
                                   uint8_t* state) {
  const uint32_t d = s->aux_size;
  uint32_t lower, upper;
  if (s->aux && s->data_type != HY_TYPE_STRING) { /* :112-124 */
    const decoded_t l = literal_of(p->value_type, &p->value), r = literal_of(p->value_type, &p->value2);
    lower = dict_bound(s, l, !lower_inclusive(p->condition));
    upper = dict_bound(s, r, upper_inclusive(p->condition));
  } else {
    if (!p->per_chunk_lower || !p->per_chunk_upper) return -1;
    lower = p->per_chunk_lower[data_chunk_id];
    upper = p->per_chunk_upper[data_chunk_id];
  }
  if (lower == 0 && upper == HY_INVALID_VALUE_ID) { /* :131-162 */
    if (p->column_is_nullable) {
      for (uint32_t i = 0; i < pos->count; ++i) {
        if (load_compressed(s->data, s->width, pos_offset(pos, i)) != d) emit(matches, n, out_chunk_id, i);
      }
    } else {
      for (uint32_t i = 0; i < pos->count; ++i) emit(matches, n, out_chunk_id, i);
      if (state) *state = HY_CHUNK_ALL_MATCH;
    }
    return 0;
  }
  if (lower == HY_INVALID_VALUE_ID || lower >= upper) { /* :167-170 */
    if (state) *state = HY_CHUNK_NONE_MATCH;
    return 0;
  }
  if (upper == HY_INVALID_VALUE_ID) upper = d; /* :178-180 */
  /* with_between_comparator(BetweenUpperExclusive, lower, upper) on ValueID (type_comparison.hpp:120-132) */
  const uint32_t lower_bound = lower, upper_bound = upper - 1;
  const uint32_t value_difference = upper_bound - lower_bound;
  for (uint32_t i = 0; i < pos->count; ++i) {
    const uint32_t vid = load_compressed(s->data, s->width, pos_offset(pos, i));
    if ((uint32_t)(vid - lower_bound) <= value_difference) emit(matches, n, out_chunk_id, i);
  }
  return 0;
}

static int scan_between_generic(const hy_segment* s, uint32_t out_chunk_id, const hy_predicate* p,
                                const positions_t* pos, hy_row_id* matches, int64_t* n) {
  if (s->data_type == HY_TYPE_STRING) return -1;
  const decoded_t l = literal_of(p->value_type, &p->value), r = literal_of(p->value_type, &p->value2);
  const int li = lower_inclusive(p->condition), ui = upper_inclusive(p->condition);
  if (!type_is_float(s->data_type)) {
    /* :86-94 empty range; then type_comparison.hpp:120-132 in the column's unsigned type */
    if (s->data_type == HY_TYPE_INT) {
      const int32_t diff = (int32_t)((uint32_t)(int32_t)r.i - (uint32_t)(int32_t)l.i) - !li - !ui;
      if (diff < 0) return 0;
      const int32_t lb = li ? (int32_t)l.i : (int32_t)l.i + 1, ub = ui ? (int32_t)r.i : (int32_t)r.i - 1;
      const uint32_t vd = (uint32_t)ub - (uint32_t)lb;
      for (uint32_t i = 0; i < pos->count; ++i) {
        const uint32_t off = pos_offset(pos, i);
        if (seg_is_null(s, off)) continue;
        if ((uint32_t)((uint32_t)(int32_t)seg_value(s, off).i - (uint32_t)lb) <= vd) emit(matches, n, out_chunk_id, i);
      }
    } else {
      const int64_t diff = (int64_t)((uint64_t)r.i - (uint64_t)l.i) - !li - !ui;
      if (diff < 0) return 0;
      const int64_t lb = li ? l.i : l.i + 1, ub = ui ? r.i : r.i - 1;
      const uint64_t vd = (uint64_t)ub - (uint64_t)lb;
      for (uint32_t i = 0; i < pos->count; ++i) {
        const uint32_t off = pos_offset(pos, i);
        if (seg_is_null(s, off)) continue;
        if ((uint64_t)((uint64_t)seg_value(s, off).i - (uint64_t)lb) <= vd) emit(matches, n, out_chunk_id, i);
      }
    }
    return 0;
  }
  for (uint32_t i = 0; i < pos->count; ++i) { /* type_comparison.hpp:135-158 */
    const uint32_t off = pos_offset(pos, i);
    if (seg_is_null(s, off)) continue;
    int match;
    if (s->data_type == HY_TYPE_FLOAT) {
      const float v = (float)seg_value(s, off).f, lo = (float)l.f, hi = (float)r.f;
      match = (li ? v >= lo : v > lo) && (ui ? v <= hi : v < hi);
    } else {
      const double v = seg_value(s, off).f;
      match = (li ? v >= l.f : v > l.f) && (ui ? v <= r.f : v < r.f);
    }
    if (match) emit(matches, n, out_chunk_id, i);
  }
  return 0;
}

/* ---- ColumnIsNull --------------------------------------------------------------------------------------------- */
static int scan_is_null(const hy_segment* s, uint32_t out_chunk_id, const hy_predicate* p, const positions_t* pos,
                        hy_row_id* matches, int64_t* n, uint8_t* state) {
  const int is_null_pred = p->condition == HY_PRED_IS_NULL;
  int all, none;
  if (s->encoding == HY_ENC_DICTIONARY) { /* _matches_all/_matches_none<BaseDictionarySegment> (:167-195) */
    all = is_null_pred ? s->aux_size == 0 : s->aux_size == s->size;
    none = is_null_pred ? s->aux_size == s->size : s->aux_size == 0;
  } else { /* BaseValueSegment (:197-223) and FrameOfReference (:225-251): nullable == has a null vector */
    all = !is_null_pred && !s->nulls;
    none = is_null_pred && !s->nulls;
  }
  if (all) { /* _add_all (:253-258) */
    for (uint32_t i = 0; i < pos->count; ++i) emit(matches, n, out_chunk_id, i);
    if (state) *state = HY_CHUNK_ALL_MATCH;
    return 0;
  }
  if (none) {
    if (state) *state = HY_CHUNK_NONE_MATCH;
    return 0;
  }
  const int invert = !is_null_pred; /* _scan_iterable_for_null_values (:153-165) */
  for (uint32_t i = 0; i < pos->count; ++i) {
    if (invert ^ seg_is_null(s, pos_offset(pos, i))) emit(matches, n, out_chunk_id, i);
  }
  return 0;
}

static int scan_non_reference_segment(const hy_segment* s, uint32_t data_chunk_id, uint32_t out_chunk_id,
                                      const hy_predicate* p, const positions_t* pos, hy_row_id* matches, int64_t* n,
                                      uint8_t* state) {
  const uint32_t c = p->condition;
  if (c == HY_PRED_IS_NULL || c == HY_PRED_IS_NOT_NULL) return scan_is_null(s, out_chunk_id, p, pos, matches, n, state);
  if (c >= HY_PRED_LIKE && c <= HY_PRED_NOT_LIKE_INSENSITIVE) {
    if (s->encoding != HY_ENC_DICTIONARY) return -1; /* unencoded strings: LikeMatcher per row, not restated */
    return scan_like_dictionary(s, data_chunk_id, out_chunk_id, p, pos, matches, n, state);
  }
  if (is_between(c)) {
    if (s->encoding == HY_ENC_DICTIONARY)
      return scan_between_dictionary(s, data_chunk_id, out_chunk_id, p, pos, matches, n, state);
    return scan_between_generic(s, out_chunk_id, p, pos, matches, n);
  }
  if (c <= HY_PRED_GREATER_THAN_EQUALS) {
    if (s->encoding == HY_ENC_DICTIONARY)
      return scan_vs_value_dictionary(s, data_chunk_id, out_chunk_id, p, pos, matches, n, state);
    return scan_vs_value_generic(s, out_chunk_id, p, pos, matches, n);
  }
  return -1;
}

int64_t hyo_scan_chunk(const hyo_column* column, uint32_t chunk_id, const hy_predicate* predicate, hy_row_id* matches,
                       uint8_t* state_out) {
  const hy_segment* seg = &column->segments[chunk_id];
  int64_t n = 0;
  if (state_out) *state_out = HY_CHUNK_SCANNED;
  if (seg->encoding != HY_ENC_REFERENCE) {
    positions_t pos = {NULL, seg->size};
    if (scan_non_reference_segment(seg, chunk_id, chunk_id, predicate, &pos, matches, &n, state_out) != 0) return -1;
    return n;
  }
  /* _scan_reference_segment (abstract_dereferenced_column_table_scan_impl.cpp:34-107) */
  const hyo_column* referenced = (const hyo_column*)seg->ref;
  const hy_row_id* pos_list = (const hy_row_id*)seg->data;
  if (seg->ref_chunk_id != 0xFFFFFFFFu && seg->size > 0) { /* fast path :38-46 */
    positions_t pos = {pos_list, seg->size}; /* pos_list == NULL: EntireChunkPosList => offset i */
    if (scan_non_reference_segment(&referenced->segments[seg->ref_chunk_id], seg->ref_chunk_id, chunk_id, predicate,
                                   &pos, matches, &n, state_out) != 0)
      return -1;
    return n;
  }
  /* slow path: split_pos_list_by_chunk_id (storage/split_pos_list_by_chunk_id.cpp:13-59) */
  const uint32_t referenced_chunks = referenced->n_chunks;
  const int include_nulls = predicate->condition == HY_PRED_IS_NULL;
  uint32_t* counts = (uint32_t*)calloc((size_t)referenced_chunks + 2, sizeof(uint32_t));
  for (uint32_t i = 0; i < seg->size; ++i) {
    if (pos_list[i].chunk_offset == 0xFFFFFFFFu) counts[referenced_chunks + 1]++;
    else counts[pos_list[i].chunk_id + 1]++;
  }
  uint32_t* starts = (uint32_t*)calloc((size_t)referenced_chunks + 2, sizeof(uint32_t));
  for (uint32_t c = 0; c <= referenced_chunks; ++c) starts[c + 1] = starts[c] + counts[c + 1];
  hy_row_id* sub_rows = (hy_row_id*)malloc(sizeof(hy_row_id) * (seg->size ? seg->size : 1));
  uint32_t* original = (uint32_t*)malloc(sizeof(uint32_t) * (seg->size ? seg->size : 1));
  uint32_t* cursor = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)referenced_chunks + 1));
  memcpy(cursor, starts, sizeof(uint32_t) * ((size_t)referenced_chunks + 1));
  for (uint32_t i = 0; i < seg->size; ++i) {
    const uint32_t bucket = pos_list[i].chunk_offset == 0xFFFFFFFFu ? referenced_chunks : pos_list[i].chunk_id;
    sub_rows[cursor[bucket]] = pos_list[i];
    original[cursor[bucket]] = i;
    cursor[bucket]++;
  }
  int rc = 0;
  for (uint32_t rc_id = 0; rc_id < referenced_chunks && rc == 0; ++rc_id) { /* :59-86 */
    const uint32_t begin = starts[rc_id], cnt = starts[rc_id + 1] - starts[rc_id];
    if (cnt == 0) continue;
    positions_t pos = {sub_rows + begin, cnt};
    const int64_t before = n;
    rc = scan_non_reference_segment(&referenced->segments[rc_id], rc_id, chunk_id, predicate, &pos, matches, &n, NULL);
    for (int64_t m = before; m < n; ++m) matches[m].chunk_offset = original[begin + matches[m].chunk_offset];
  }
  if (rc == 0 && include_nulls) { /* :92-106 */
    const uint32_t begin = starts[referenced_chunks], cnt = starts[referenced_chunks + 1] - begin;
    for (uint32_t i = 0; i < cnt; ++i) emit(matches, &n, chunk_id, original[begin + i]);
  }
  free(counts); free(starts); free(sub_rows); free(original); free(cursor);
  if (state_out) *state_out = HY_CHUNK_SCANNED;
  return rc == 0 ? n : -1;
}

/* ---- ColumnVsColumn ------------------------------------------------------------------------------------------- */
typedef struct {
  int is_null;
  uint32_t type;
  decoded_t v;
} cell_t;

static cell_t column_cell(const hyo_column* col, uint32_t chunk_id, uint32_t i) {
  cell_t c;
  const hy_segment* s = &col->segments[chunk_id];
  c.type = s->data_type;
  c.v.i = 0; c.v.f = 0.0;
  if (s->encoding == HY_ENC_REFERENCE) { /* ReferenceSegmentIterable: NULL_ROW_ID => NULL */
    const hyo_column* referenced = (const hyo_column*)s->ref;
    hy_row_id r;
    if (s->data) r = ((const hy_row_id*)s->data)[i];
    else { r.chunk_id = s->ref_chunk_id; r.chunk_offset = i; }
    if (r.chunk_offset == 0xFFFFFFFFu) { c.is_null = 1; return c; }
    const hy_segment* d = &referenced->segments[r.chunk_id];
    c.is_null = seg_is_null(d, r.chunk_offset);
    if (!c.is_null) c.v = seg_value(d, r.chunk_offset);
    return c;
  }
  c.is_null = seg_is_null(s, i);
  if (!c.is_null) c.v = seg_value(s, i);
  return c;
}

/* C++ usual arithmetic conversions of `left OP right` for the four numeric column types. */
static int compare_cells(uint32_t cond, const cell_t* l, const cell_t* r) {
  const int lf = type_is_float(l->type), rf = type_is_float(r->type);
  if (!lf && !rf) return cmp_i64(cond, l->v.i, r->v.i);
  if (l->type == HY_TYPE_DOUBLE || r->type == HY_TYPE_DOUBLE) {
    const double a = lf ? l->v.f : (double)l->v.i, b = rf ? r->v.f : (double)r->v.i;
    return cmp_f64(cond, a, b);
  }
  const float a = lf ? (float)l->v.f : (float)l->v.i, b = rf ? (float)r->v.f : (float)r->v.i;
  return cmp_f32(cond, a, b);
}

int64_t hyo_scan_chunk_columns(const hyo_column* left, const hyo_column* right, uint32_t chunk_id, uint32_t condition,
                               hy_row_id* matches) {
  const hy_segment* ls = &left->segments[chunk_id];
  const hy_segment* rs = &right->segments[chunk_id];
  if (ls->data_type == HY_TYPE_STRING || rs->data_type == HY_TYPE_STRING) return -1;
  if (condition > HY_PRED_GREATER_THAN_EQUALS || ls->size != rs->size) return -1;
  int64_t n = 0;
  for (uint32_t i = 0; i < ls->size; ++i) { /* _scan_with_iterators<true>(cmp, left, right) :169-184 */
    const cell_t l = column_cell(left, chunk_id, i), r = column_cell(right, chunk_id, i);
    if (l.is_null || r.is_null) continue;
    if (compare_cells(condition, &l, &r)) emit(matches, &n, chunk_id, i);
  }
  return n;
}

/* ---- whole-column drivers (TableScan::_on_execute's fan-out, table_scan.cpp:119-232) --------------------------- */
typedef struct {
  const hyo_column* column;
  const hyo_column* right;
  const hy_predicate* predicate;
  uint32_t condition;
  hy_scan_result* result;
  const uint64_t* region; /* start of each chunk's private region in result->matches */
  uint32_t begin, end;
  int rc;
} scan_job_t;

static void* scan_job_run(void* arg) {
  scan_job_t* j = (scan_job_t*)arg;
  for (uint32_t c = j->begin; c < j->end; ++c) {
    uint8_t state = HY_CHUNK_SCANNED;
    hy_row_id* out = j->result->matches + j->region[c];
    const int64_t n = j->right ? hyo_scan_chunk_columns(j->column, j->right, c, j->condition, out)
                               : hyo_scan_chunk(j->column, c, j->predicate, out, &state);
    if (n < 0) { j->rc = -1; return NULL; }
    j->result->counts[c] = (uint32_t)n;
    j->result->chunk_state[c] = state;
  }
  return NULL;
}

static int32_t run_scan(const hyo_column* column, const hyo_column* right, const hy_predicate* predicate,
                        uint32_t condition, hy_scan_result* result, int threads) {
  const uint32_t n_chunks = column->n_chunks;
  uint64_t* region = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n_chunks + 1));
  region[0] = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) region[c + 1] = region[c] + column->segments[c].size;
  if (region[n_chunks] > result->capacity) { free(region); return HY_ERR_CAPACITY; }
  if (threads < 1) threads = 1;
  if ((uint32_t)threads > n_chunks) threads = n_chunks ? (int)n_chunks : 1;
  scan_job_t* jobs = (scan_job_t*)calloc((size_t)threads, sizeof(scan_job_t));
  pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    jobs[t].column = column; jobs[t].right = right; jobs[t].predicate = predicate; jobs[t].condition = condition;
    jobs[t].result = result; jobs[t].region = region;
    jobs[t].begin = (uint32_t)((uint64_t)n_chunks * (uint64_t)t / (uint64_t)threads);
    jobs[t].end = (uint32_t)((uint64_t)n_chunks * (uint64_t)(t + 1) / (uint64_t)threads);
    if (threads == 1) scan_job_run(&jobs[t]);
    else pthread_create(&tids[t], NULL, scan_job_run, &jobs[t]);
  }
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    if (threads > 1) pthread_join(tids[t], NULL);
    rc |= jobs[t].rc;
  }
  free(jobs); free(tids);
  if (rc) { free(region); return HY_ERR_UNSUPPORTED; }
  /* Compact the per-chunk regions into the ABI's back-to-back layout; ALL_MATCH chunks keep count == size but
   * own no RowIDs unless HY_SCAN_MATERIALIZE_ALL_MATCH is set. */
  uint64_t cursor = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    result->offsets[c] = cursor;
    const int elide = result->chunk_state[c] == HY_CHUNK_ALL_MATCH && !(result->flags & HY_SCAN_MATERIALIZE_ALL_MATCH);
    if (!elide) {
      memmove(result->matches + cursor, result->matches + region[c], sizeof(hy_row_id) * result->counts[c]);
      cursor += result->counts[c];
    }
  }
  result->offsets[n_chunks] = cursor;
  result->total_matches = cursor;
  free(region);
  return HY_OK;
}

int32_t hyo_table_scan(const hyo_column* column, const hy_predicate* predicate, hy_scan_result* result, int threads) {
  return run_scan(column, NULL, predicate, 0, result, threads);
}

int32_t hyo_table_scan_columns(const hyo_column* left, const hyo_column* right, uint32_t condition,
                               hy_scan_result* result, int threads) {
  if (left->n_chunks != right->n_chunks) return HY_ERR_INVALID;
  return run_scan(left, right, NULL, condition, result, threads);
}
