/*
 * aggregate.c -- restatement of AggregateHash.  TEST INFRASTRUCTURE ONLY (see hy_oracle.h).
 *
 * Follows:
 *   AggregateHash::_partition_by_groupby_keys        operators/aggregate_hash.cpp:661-948
 *   get_or_add_result (first-occurrence result ids,
 *     immediate-key shortcut)                        aggregate_hash.cpp:317-403, 770-804
 *   AggregateHash::_aggregate / _aggregate_segment   aggregate_hash.cpp:605-655, 950-1178
 *   WindowFunctionBuilder (accumulators)             operators/abstract_aggregate_operator.hpp:30-133
 *   WindowFunctionTraits (result types)              operators/aggregate/window_function_traits.hpp:11-77
 *   write_aggregate_values / write_groupby_output    aggregate_hash.cpp:57-221, 421-537
 * The reference aggregates strictly in row order on one thread, so SUM/AVG over float/double columns are reproducible
 * double additions; this file does the same and is therefore the bit-level oracle for them.
 *
 * Group identity is the tuple of (is NULL, value) over the GROUP BY columns -- the numeric AggregateKeyEntry the
 * reference derives (int32: value - INT32_MIN + 1, short strings: byte packing, ...) is only a name for it.  String
 * GROUP BY columns reach this layer (and the device ABI) as dictionary segments whose "dictionary" holds that name
 * (an int64 per value id), see INTEGRATION.md.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "hy_oracle.h"

#define MAX_GROUPBY 16

static inline uint32_t load_compressed(const void* data, uint32_t width, uint32_t i) {
  if (width == 1) return ((const uint8_t*)data)[i];
  if (width == 2) return ((const uint16_t*)data)[i];
  return ((const uint32_t*)data)[i];
}

typedef struct {
  int is_null;
  int64_t i;
  double f;
} cell_t;

static cell_t data_cell(const hy_segment* s, uint32_t row) {
  cell_t c = {0, 0, 0.0};
  const void* values = s->data;
  uint32_t index = row;
  if (s->encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = load_compressed(s->data, s->width, row);
    if (vid >= s->aux_size) { c.is_null = 1; return c; }
    values = s->aux;
    index = vid;
  } else {
    if (s->nulls && ((s->nulls[row / 64] >> (row % 64)) & 1)) { c.is_null = 1; return c; }
    if (s->encoding == HY_ENC_FRAME_OF_REFERENCE) {
      c.i = (int32_t)(load_compressed(s->data, s->width, row) + (uint32_t)((const int32_t*)s->aux)[row / HY_FOR_BLOCK_SIZE]);
      return c;
    }
  }
  switch (s->data_type) {
    case HY_TYPE_INT: c.i = ((const int32_t*)values)[index]; break;
    case HY_TYPE_LONG: c.i = ((const int64_t*)values)[index]; break;
    case HY_TYPE_FLOAT: c.f = ((const float*)values)[index]; break;
    case HY_TYPE_DOUBLE: c.f = ((const double*)values)[index]; break;
    default: c.is_null = 1; break;
  }
  return c;
}

static cell_t column_cell(const hyo_column* col, uint32_t chunk, uint32_t row) {
  const hy_segment* s = &col->segments[chunk];
  if (s->encoding != HY_ENC_REFERENCE) return data_cell(s, row);
  const hyo_column* referenced = (const hyo_column*)s->ref;
  hy_row_id r;
  if (s->data) r = ((const hy_row_id*)s->data)[row];
  else { r.chunk_id = s->ref_chunk_id; r.chunk_offset = row; }
  if (r.chunk_offset == 0xFFFFFFFFu) { cell_t c = {1, 0, 0.0}; return c; }
  return data_cell(&referenced->segments[r.chunk_id], r.chunk_offset);
}

static int is_float_type(uint32_t t) { return t == HY_TYPE_FLOAT || t == HY_TYPE_DOUBLE; }

typedef struct {
  uint32_t null_mask;
  int64_t keys[MAX_GROUPBY];
} group_key_t;

typedef struct {
  /* per aggregate */
  double f;            /* SUM/AVG (double), MIN/MAX of float/double */
  int64_t i;           /* SUM (int64), MIN/MAX of int */
  uint64_t count;      /* aggregate_count: non-NULL inputs (COUNT(*): rows) */
  double welford[4];   /* STDDEV_SAMP: count, mean, M2, result (abstract_aggregate_operator.hpp:83-113) */
  int64_t* distinct;   /* COUNT DISTINCT: value bits seen */
  uint64_t n_distinct, cap_distinct;
} accumulator_t;

typedef struct {
  group_key_t key;
  hy_row_id first_row, last_row;
  accumulator_t* acc;
} group_t;

static uint64_t hash_key(const group_key_t* k, uint32_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ULL ^ k->null_mask;
  for (uint32_t c = 0; c < n; ++c) {
    h ^= (uint64_t)k->keys[c] + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
  }
  return h;
}

static uint32_t result_type(uint32_t function, uint32_t input_type) { /* window_function_traits.hpp */
  switch (function) {
    case HY_AGG_COUNT:
    case HY_AGG_COUNT_DISTINCT: return HY_TYPE_LONG;
    case HY_AGG_AVG:
    case HY_AGG_STDDEV_SAMP: return HY_TYPE_DOUBLE;
    case HY_AGG_SUM: return is_float_type(input_type) ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
    default: return input_type; /* MIN / MAX / ANY */
  }
}

int32_t hyo_aggregate_hash(const hyo_column* const* groupby_columns, uint32_t n_groupby, const uint32_t* functions,
                           const hyo_column* const* aggregate_columns, uint32_t n_aggregates,
                           hy_aggregate_result* result) {
  if (n_groupby > MAX_GROUPBY) return HY_ERR_UNSUPPORTED;
  const hyo_column* shape = n_groupby ? groupby_columns[0] : NULL;
  for (uint32_t a = 0; a < n_aggregates && !shape; ++a) shape = aggregate_columns[a];
  if (!shape) return HY_ERR_INVALID; /* the caller passes at least one column to define the table's chunks */
  uint64_t total_rows = 0;
  for (uint32_t c = 0; c < shape->n_chunks; ++c) total_rows += shape->segments[c].size;

  uint64_t cap = 1024;
  while (cap < total_rows * 2) cap <<= 1;
  uint32_t* table = (uint32_t*)malloc(sizeof(uint32_t) * cap);
  memset(table, 0xFF, sizeof(uint32_t) * cap);
  group_t* groups = NULL;
  uint64_t n_groups = 0, cap_groups = 0;

  /* row order: chunk by chunk, row by row (aggregate_hash.cpp:1016-1176) */
  for (uint32_t chunk = 0; chunk < shape->n_chunks; ++chunk) {
    const uint32_t rows = shape->segments[chunk].size;
    for (uint32_t row = 0; row < rows; ++row) {
      group_key_t key;
      memset(&key, 0, sizeof(key));
      for (uint32_t g = 0; g < n_groupby; ++g) {
        const cell_t v = column_cell(groupby_columns[g], chunk, row);
        const uint32_t t = groupby_columns[g]->segments[chunk].data_type;
        if (v.is_null) key.null_mask |= 1u << g;
        else if (is_float_type(t)) { double d = v.f; if (t == HY_TYPE_FLOAT) d = (float)v.f; memcpy(&key.keys[g], &d, 8); if (d == 0.0) key.keys[g] = 0; }
        else key.keys[g] = v.i;
      }
      uint64_t slot = hash_key(&key, n_groupby) & (cap - 1);
      uint32_t id;
      for (;;) {
        id = table[slot];
        if (id == 0xFFFFFFFFu) break;
        if (groups[id].key.null_mask == key.null_mask && memcmp(groups[id].key.keys, key.keys, sizeof(int64_t) * n_groupby) == 0) break;
        slot = (slot + 1) & (cap - 1);
      }
      if (id == 0xFFFFFFFFu) { /* first occurrence: new result id (aggregate_hash.cpp:388-401) */
        if (n_groups == cap_groups) {
          cap_groups = cap_groups ? cap_groups * 2 : 64;
          groups = (group_t*)realloc(groups, sizeof(group_t) * cap_groups);
        }
        id = (uint32_t)n_groups++;
        table[slot] = id;
        groups[id].key = key;
        groups[id].first_row.chunk_id = chunk;
        groups[id].first_row.chunk_offset = row;
        groups[id].acc = (accumulator_t*)calloc(n_aggregates ? n_aggregates : 1, sizeof(accumulator_t));
      }
      group_t* grp = &groups[id];
      grp->last_row.chunk_id = chunk;
      grp->last_row.chunk_offset = row;
      for (uint32_t a = 0; a < n_aggregates; ++a) {
        accumulator_t* acc = &grp->acc[a];
        if (!aggregate_columns[a]) { acc->count++; continue; } /* COUNT(*) (:1075-1119) */
        const cell_t v = column_cell(aggregate_columns[a], chunk, row);
        if (v.is_null) continue;                                 /* :627-637 */
        const uint32_t t = aggregate_columns[a]->segments[chunk].data_type;
        const int fl = is_float_type(t);
        const double as_double = fl ? (t == HY_TYPE_FLOAT ? (double)(float)v.f : v.f) : (double)v.i;
        switch (functions[a]) {
          case HY_AGG_MIN:
            if (fl) { if (acc->count == 0 || as_double < acc->f) acc->f = as_double; }
            else if (acc->count == 0 || v.i < acc->i) acc->i = v.i;
            break;
          case HY_AGG_MAX:
            if (fl) { if (acc->count == 0 || as_double > acc->f) acc->f = as_double; }
            else if (acc->count == 0 || v.i > acc->i) acc->i = v.i;
            break;
          case HY_AGG_SUM:
          case HY_AGG_AVG:
            if (fl || functions[a] == HY_AGG_AVG) acc->f += as_double; /* AVG accumulates in double for every type */
            if (!fl) acc->i += v.i;
            break;
          case HY_AGG_STDDEV_SAMP: {
            double* w = acc->welford;
            w[0] += 1.0;
            const double delta = as_double - w[1];
            w[1] += delta / w[0];
            const double delta2 = as_double - w[1];
            w[2] += delta * delta2;
            if (w[0] > 1.0) w[3] = sqrt(w[2] / (w[0] - 1.0));
            break;
          }
          case HY_AGG_COUNT_DISTINCT: {
            int64_t bits = v.i;
            if (fl) { double d = as_double; if (d == 0.0) d = 0.0; memcpy(&bits, &d, 8); }
            int seen = 0;
            for (uint64_t k = 0; k < acc->n_distinct && !seen; ++k) seen = acc->distinct[k] == bits;
            if (!seen) {
              if (acc->n_distinct == acc->cap_distinct) {
                acc->cap_distinct = acc->cap_distinct ? acc->cap_distinct * 2 : 8;
                acc->distinct = (int64_t*)realloc(acc->distinct, sizeof(int64_t) * acc->cap_distinct);
              }
              acc->distinct[acc->n_distinct++] = bits;
            }
            break;
          }
          default: break; /* COUNT: only the counter; ANY: resolved from the representative row */
        }
        acc->count++;
      }
    }
  }

  /* Immediate-key shortcut (aggregate_hash.cpp:770-804): ONE int32 GROUP BY column whose key range is below
   * 1.2 x rows => results are indexed by key (NULL first), and the representative row is the LAST row of the group. */
  int immediate = 0;
  if (n_groupby == 1 && groupby_columns[0]->n_chunks > 0 && groupby_columns[0]->segments[0].data_type == HY_TYPE_INT) {
    uint64_t min_key = UINT64_MAX, max_key = 0;
    for (uint64_t g = 0; g < n_groups; ++g) {
      if (groups[g].key.null_mask) continue;
      const uint64_t k = (uint64_t)(groups[g].key.keys[0] - (int64_t)INT32_MIN) + 1;
      if (k < min_key) min_key = k;
      if (k > max_key) max_key = k;
    }
    if (max_key > 0 && (double)(max_key - min_key) < (double)total_rows * 1.2) immediate = 1;
  }
  uint64_t* order = (uint64_t*)malloc(sizeof(uint64_t) * (n_groups ? n_groups : 1));
  for (uint64_t g = 0; g < n_groups; ++g) order[g] = g;
  if (immediate) { /* ascending key, NULL (index 0) first; insertion sort is fine for test sizes, else qsort-like */
    for (uint64_t i = 1; i < n_groups; ++i) {
      const uint64_t id = order[i];
      const int id_null = groups[id].key.null_mask != 0;
      uint64_t j = i;
      while (j > 0) {
        const uint64_t other = order[j - 1];
        const int other_null = groups[other].key.null_mask != 0;
        const int other_greater = id_null ? !other_null : (!other_null && groups[other].key.keys[0] > groups[id].key.keys[0]);
        if (!other_greater) break;
        order[j] = other;
        --j;
      }
      order[j] = id;
    }
  }

  int32_t status = HY_OK;
  uint64_t out_groups = n_groups;
  const int no_groupby_empty = n_groupby == 0 && n_groups == 0; /* one row: NULL / 0 (aggregate_hash.cpp:1422-1432) */
  if (no_groupby_empty) out_groups = 1;
  if (out_groups > result->group_capacity) status = HY_ERR_CAPACITY;
  else {
    result->n_groups = (uint32_t)out_groups;
    for (uint64_t o = 0; o < out_groups; ++o) {
      const group_t* grp = no_groupby_empty ? NULL : &groups[order[o]];
      if (result->group_row_ids) {
        hy_row_id rid = {0, 0};
        if (grp) rid = immediate ? grp->last_row : grp->first_row;
        result->group_row_ids[o] = rid;
      }
      for (uint32_t a = 0; a < n_aggregates; ++a) {
        hy_aggregate_column* col = &result->columns[a];
        const uint32_t in_type = aggregate_columns[a] && aggregate_columns[a]->n_chunks ? aggregate_columns[a]->segments[0].data_type : HY_TYPE_LONG;
        const uint32_t out_type = result_type(functions[a], in_type);
        col->data_type = out_type;
        const accumulator_t* acc = grp ? &grp->acc[a] : NULL;
        const uint64_t count = acc ? acc->count : 0;
        int is_null = 0;
        int64_t vi = 0;
        double vf = 0.0;
        switch (functions[a]) {
          case HY_AGG_COUNT: vi = (int64_t)count; break;
          case HY_AGG_COUNT_DISTINCT: vi = acc ? (int64_t)acc->n_distinct : 0; break;
          case HY_AGG_SUM: is_null = count == 0; if (acc) { vi = acc->i; vf = acc->f; } break;
          case HY_AGG_AVG: is_null = count == 0; if (acc && count) vf = acc->f / (double)count; break; /* :166 */
          case HY_AGG_STDDEV_SAMP: is_null = count <= 1; if (acc && count > 1) vf = acc->welford[3]; break;
          case HY_AGG_MIN:
          case HY_AGG_MAX: is_null = count == 0; if (acc) { vi = acc->i; vf = acc->f; } break;
          case HY_AGG_ANY: {
            if (grp && aggregate_columns[a]) {
              const hy_row_id rid = immediate ? grp->last_row : grp->first_row;
              const cell_t v = column_cell(aggregate_columns[a], rid.chunk_id, rid.chunk_offset);
              is_null = v.is_null; vi = v.i; vf = v.f;
            } else is_null = 1;
            break;
          }
          default: status = HY_ERR_UNSUPPORTED; break;
        }
        if (col->is_null) col->is_null[o] = (uint8_t)is_null;
        switch (out_type) {
          case HY_TYPE_INT: ((int32_t*)col->values)[o] = is_null ? 0 : (int32_t)vi; break;
          case HY_TYPE_LONG: ((int64_t*)col->values)[o] = is_null ? 0 : vi; break;
          case HY_TYPE_FLOAT: ((float*)col->values)[o] = is_null ? 0.f : (float)vf; break;
          default: ((double*)col->values)[o] = is_null ? 0.0 : vf; break;
        }
      }
    }
  }
  for (uint64_t g = 0; g < n_groups; ++g) {
    for (uint32_t a = 0; a < n_aggregates; ++a) free(groups[g].acc[a].distinct);
    free(groups[g].acc);
  }
  free(groups); free(table); free(order);
  return status;
}
