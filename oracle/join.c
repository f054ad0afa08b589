/*
 * join.c -- restatement of JoinHash (radix-partitioned hash join).  TEST INFRASTRUCTURE ONLY (see hy_oracle.h).
 *
 * Follows, step by step:
 *   JoinHash::calculate_radix_bits                    operators/join_hash.cpp:70-114
 *   JoinHash::_on_execute (side selection)            join_hash.cpp:139-183
 *   JoinHashImpl::_on_execute (NULL policy, Bloom
 *     filter order, AntiNullAsTrue early exit)        join_hash.cpp:270-572
 *   materialize_input                                 join_hash/join_hash_steps.hpp:274-420
 *   partition_by_radix                                join_hash_steps.hpp:509-617
 *   build + PosHashTable                              join_hash_steps.hpp:97-236, 426-507
 *   probe / probe_semi_anti                           join_hash_steps.hpp:624-922
 * Keys: int32/int64 -- std::hash is the identity there (pinned by join_hash_steps_test.cpp:169-188) -- and float/double,
 * also mixed with integers (JoinHashTraits, join_hash_traits.hpp:15-40: both sides are cast to one HashedType first).
 * std::hash<float/double> is implementation-defined; this file restates libstdc++'s (GCC is the reference's primary
 * compiler): 0 for +-0.0, otherwise _Hash_bytes (libsupc++/hash_bytes.cc, the 64-bit Murmur-style function, seed
 * 0xc70f6907) over the value's 4 / 8 bytes.  tests/test_oracle_join.py pins hyo_std_hash against the std::hash of the g++
 * installed here.  String keys stay on the CPU path.
 * Secondary predicates (MultiPredicateJoinEvaluator, join_hash_steps.hpp:563-591: every candidate pair is tested against the
 * additional column comparisons before it is emitted) are restated below (hyo_join_hash_predicates).
 *
 * Result: the concatenation of probe()'s per-slice PosLists, in the order the slices are created
 * (partition, then 131 070-element slice), plus the slice boundaries; write_output_chunks
 * (join_helper/join_output_writing.cpp:205-340) only groups / merges slices and is host-side bookkeeping.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "hy_oracle.h"

#define BLOOM_BITS (1u << 20)       /* join_hash_steps.hpp:252 */
#define BLOOM_MASK (BLOOM_BITS - 1)
#define BLOOM_WORDS (BLOOM_BITS / 64)
#define PROBE_SIZE_PER_CHUNK (65535u * 2u) /* join_hash_steps.hpp:47 */

typedef struct {
  hy_row_id row_id;
  int64_t value;
} element_t;

typedef struct {
  element_t* elements;
  uint8_t* nulls; /* byte per element, only with keep_nulls */
  uint64_t size;
} partition_t;

static inline uint32_t load_compressed(const void* data, uint32_t width, uint32_t i) {
  if (width == 1) return ((const uint8_t*)data)[i];
  if (width == 2) return ((const uint16_t*)data)[i];
  return ((const uint32_t*)data)[i];
}

/* value of row i of a DATA segment; returns 1 if NULL */
static int data_value(const hy_segment* s, uint32_t i, int64_t* out) {
  *out = 0;
  switch (s->encoding) {
    case HY_ENC_UNENCODED:
      if (s->nulls && ((s->nulls[i / 64] >> (i % 64)) & 1)) return 1;
      *out = s->data_type == HY_TYPE_INT ? ((const int32_t*)s->data)[i] : ((const int64_t*)s->data)[i];
      return 0;
    case HY_ENC_DICTIONARY: {
      const uint32_t vid = load_compressed(s->data, s->width, i);
      if (vid >= s->aux_size) return 1;
      *out = s->data_type == HY_TYPE_INT ? ((const int32_t*)s->aux)[vid] : ((const int64_t*)s->aux)[vid];
      return 0;
    }
    case HY_ENC_FRAME_OF_REFERENCE:
      if (s->nulls && ((s->nulls[i / 64] >> (i % 64)) & 1)) return 1;
      *out = (int32_t)(load_compressed(s->data, s->width, i) + (uint32_t)((const int32_t*)s->aux)[i / HY_FOR_BLOCK_SIZE]);
      return 0;
    default: return 1;
  }
}

/* value of row i of chunk c of a column that may be made of reference segments */
static int column_value(const hyo_column* col, uint32_t c, uint32_t i, int64_t* out) {
  const hy_segment* s = &col->segments[c];
  if (s->encoding != HY_ENC_REFERENCE) return data_value(s, i, out);
  const hyo_column* referenced = (const hyo_column*)s->ref;
  hy_row_id r;
  if (s->data) r = ((const hy_row_id*)s->data)[i];
  else { r.chunk_id = s->ref_chunk_id; r.chunk_offset = i; }
  *out = 0;
  if (r.chunk_offset == 0xFFFFFFFFu) return 1;
  return data_value(&referenced->segments[r.chunk_id], r.chunk_offset, out);
}

/* ---- secondary predicates: MultiPredicateJoinEvaluator (multi_predicate_join_evaluator.hpp:30-61) ----------------------- */
typedef struct { int is_null; int64_t i; double f; } typed_value_t;   /* .i for int / long, .f for float (widened) / double */

static typed_value_t data_typed_value(const hy_segment* s, uint32_t i) {
  typed_value_t v = {0, 0, 0.0};
  const void* values = s->data;
  uint32_t index = i;
  if (s->encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = load_compressed(s->data, s->width, i);
    if (vid >= s->aux_size) { v.is_null = 1; return v; }
    values = s->aux;
    index = vid;
  } else {
    if (s->nulls && ((s->nulls[i / 64] >> (i % 64)) & 1)) { v.is_null = 1; return v; }
    if (s->encoding == HY_ENC_FRAME_OF_REFERENCE) {
      v.i = (int32_t)(load_compressed(s->data, s->width, i) + (uint32_t)((const int32_t*)s->aux)[i / HY_FOR_BLOCK_SIZE]);
      return v;
    }
  }
  switch (s->data_type) {
    case HY_TYPE_INT: v.i = ((const int32_t*)values)[index]; break;
    case HY_TYPE_LONG: v.i = ((const int64_t*)values)[index]; break;
    case HY_TYPE_FLOAT: v.f = ((const float*)values)[index]; break;
    default: v.f = ((const double*)values)[index]; break;
  }
  return v;
}

static typed_value_t column_typed_value(const hyo_column* col, hy_row_id id, uint32_t* type) {
  const hy_segment* s = &col->segments[id.chunk_id];
  *type = s->data_type;
  if (s->encoding != HY_ENC_REFERENCE) return data_typed_value(s, id.chunk_offset);
  const hyo_column* referenced = (const hyo_column*)s->ref;
  hy_row_id r;
  if (s->data) r = ((const hy_row_id*)s->data)[id.chunk_offset];
  else { r.chunk_id = s->ref_chunk_id; r.chunk_offset = id.chunk_offset; }
  typed_value_t null_value = {1, 0, 0.0};
  if (r.chunk_offset == 0xFFFFFFFFu) return null_value;
  return data_typed_value(&referenced->segments[r.chunk_id], r.chunk_offset);
}

static int is_float_type(uint32_t t) { return t == HY_TYPE_FLOAT || t == HY_TYPE_DOUBLE; }

/* x <condition> y in the common C++ type of the two column types (the comparator functors of type_comparison.hpp are
 * generic lambdas: the usual arithmetic conversions apply -- int64 against float compares as float) */
static int compare_values(uint32_t condition, typed_value_t x, uint32_t xt, typed_value_t y, uint32_t yt) {
  int less, equal;
  if (xt == HY_TYPE_DOUBLE || yt == HY_TYPE_DOUBLE) {
    const double a = is_float_type(xt) ? x.f : (double)x.i, b = is_float_type(yt) ? y.f : (double)y.i;
    less = a < b; equal = a == b;
  } else if (xt == HY_TYPE_FLOAT || yt == HY_TYPE_FLOAT) {
    const float a = is_float_type(xt) ? (float)x.f : (float)x.i, b = is_float_type(yt) ? (float)y.f : (float)y.i;
    less = a < b; equal = a == b;
  } else {
    less = x.i < y.i; equal = x.i == y.i;
  }
  switch (condition) {
    case HY_PRED_EQUALS: return equal;
    case HY_PRED_NOT_EQUALS: return !equal;
    case HY_PRED_LESS_THAN: return less;
    case HY_PRED_LESS_THAN_EQUALS: return less || equal;
    case HY_PRED_GREATER_THAN: return !less && !equal;
    case HY_PRED_GREATER_THAN_EQUALS: return !less;
    default: return 0;
  }
}

typedef struct { const hyo_column* build; const hyo_column* probe; uint32_t condition; } secondary_t;   /* build <condition> probe */

static int satisfies_all_predicates(const secondary_t* predicates, uint32_t n, hy_row_id build_row, hy_row_id probe_row) {
  for (uint32_t p = 0; p < n; ++p) {
    uint32_t bt, pt;
    const typed_value_t b = column_typed_value(predicates[p].build, build_row, &bt), q = column_typed_value(predicates[p].probe, probe_row, &pt);
    if (b.is_null || q.is_null) return 0;   /* :50-52 (AntiNullAsTrue is not supported with secondary predicates) */
    if (!compare_values(predicates[p].condition, b, bt, q, pt)) return 0;
  }
  return 1;
}

static uint32_t flip_condition(uint32_t c) {   /* flip_predicate_condition, types.cpp */
  switch (c) {
    case HY_PRED_LESS_THAN: return HY_PRED_GREATER_THAN;
    case HY_PRED_LESS_THAN_EQUALS: return HY_PRED_GREATER_THAN_EQUALS;
    case HY_PRED_GREATER_THAN: return HY_PRED_LESS_THAN;
    case HY_PRED_GREATER_THAN_EQUALS: return HY_PRED_LESS_THAN_EQUALS;
    default: return c;
  }
}

static inline int bloom_get(const uint64_t* bloom, uint64_t hash) {
  const uint32_t bit = (uint32_t)(hash & BLOOM_MASK);
  return (int)((bloom[bit / 64] >> (bit % 64)) & 1);
}
static inline void bloom_set(uint64_t* bloom, uint64_t hash) {
  const uint32_t bit = (uint32_t)(hash & BLOOM_MASK);
  bloom[bit / 64] |= (uint64_t)1 << (bit % 64);
}

uint32_t hyo_calculate_radix_bits(uint64_t build_rows, uint64_t probe_rows) {
  (void)probe_rows;
  const double l2_cache_max_usable = 1024000 * 0.75;                             /* join_hash.cpp:95-96 */
  const double complete_hash_map_size = (double)build_rows * (double)sizeof(uint32_t) / 0.8; /* :101-105 */
  double cluster_count = complete_hash_map_size / l2_cache_max_usable;
  if (cluster_count < 1.0) cluster_count = 1.0;
  const double bits = ceil(log2(cluster_count));
  return bits > 8.0 ? 8u : (uint32_t)bits;                                       /* :113 */
}

/* ---- HashedType (join_hash_traits.hpp:15-40) and its std::hash ------------------------------------------------------------- */
/* One join at a time (test infrastructure): the type both sides are cast to before hashing and comparing.  Integer joins keep
 * the element's value; float joins keep the bit pattern of the HashedType value (float: zero-extended), -0.0 as +0.0 (they
 * compare equal and both hash to 0), so that equal bits <=> equal keys -- except NaN, which equals nothing. */
static uint32_t g_hashed_type = HY_TYPE_LONG;

static uint32_t hashed_type_of(uint32_t l, uint32_t r) {
  const int lf = is_float_type(l), rf = is_float_type(r);
  if (lf && rf) return (l == HY_TYPE_DOUBLE || r == HY_TYPE_DOUBLE) ? HY_TYPE_DOUBLE : HY_TYPE_FLOAT;   /* the larger one */
  if (!lf && !rf) return (l == HY_TYPE_LONG || r == HY_TYPE_LONG) ? HY_TYPE_LONG : HY_TYPE_INT;
  return lf ? l : r;                                                                                   /* the floating one */
}

uint32_t hyo_join_hashed_type(uint32_t left_type, uint32_t right_type) { return hashed_type_of(left_type, right_type); }   /* (tests: join_hash_traits_test.cpp) */

/* libstdc++ _Hash_bytes, size_t = 64 bits (libsupc++/hash_bytes.cc) */
static uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }
static uint64_t libstdcxx_hash_bytes(const void* ptr, uint64_t len) {
  const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
  const unsigned char* buf = (const unsigned char*)ptr;
  const uint64_t len_aligned = len & ~(uint64_t)7;
  uint64_t hash = (uint64_t)0xc70f6907UL ^ (len * mul);
  for (uint64_t p = 0; p < len_aligned; p += 8) {
    uint64_t word;
    memcpy(&word, buf + p, 8);
    hash ^= shift_mix(word * mul) * mul;
    hash *= mul;
  }
  if (len & 7) {
    uint64_t data = 0;
    for (int n = (int)(len & 7) - 1; n >= 0; --n) data = (data << 8) + buf[len_aligned + (uint64_t)n];
    hash ^= data;
    hash *= mul;
  }
  hash = shift_mix(hash) * mul;
  return shift_mix(hash);
}

uint64_t hyo_std_hash_bytes(const void* data, uint64_t length) { return libstdcxx_hash_bytes(data, length); }   /* std::hash<std::string> */

/* std::hash<HashedType>{}(key) for a key in this file's representation (functional_hash.h: 0 for 0.0 and -0.0) */
uint64_t hyo_std_hash(int64_t key, uint32_t hashed_type) {
  if (hashed_type == HY_TYPE_FLOAT) {
    const uint32_t bits = (uint32_t)key;
    return bits == 0 ? 0 : libstdcxx_hash_bytes(&bits, 4);
  }
  if (hashed_type == HY_TYPE_DOUBLE) return key == 0 ? 0 : libstdcxx_hash_bytes(&key, 8);
  return (uint64_t)key;   /* std::hash<integral> is the identity */
}

static int key_is_nan(int64_t key) {
  if (g_hashed_type == HY_TYPE_FLOAT) return ((uint32_t)key & 0x7FFFFFFFu) > 0x7F800000u;
  if (g_hashed_type == HY_TYPE_DOUBLE) return ((uint64_t)key & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull;
  return 0;
}

/* static_cast<HashedType>(value) in this file's key representation */
static int64_t key_of(typed_value_t v, uint32_t column_type) {
  if (g_hashed_type == HY_TYPE_FLOAT) {
    const float f = is_float_type(column_type) ? (float)v.f : (float)v.i;
    uint32_t bits;
    memcpy(&bits, &f, 4);
    return f == 0.0f ? 0 : (int64_t)bits;
  }
  if (g_hashed_type == HY_TYPE_DOUBLE) {
    const double d = is_float_type(column_type) ? v.f : (double)v.i;
    int64_t bits;
    memcpy(&bits, &d, 8);
    return d == 0.0 ? 0 : bits;
  }
  return v.i;
}

/* materialize_input<T, HashedType, keep_null_values> for one chunk (join_hash_steps.hpp:306-410). */
static void materialize_chunk(const hyo_column* col, uint32_t chunk_id, int keep_nulls, uint32_t radix_bits,
                              const uint64_t* bloom_in, uint64_t* bloom_out, partition_t* out, uint64_t* histogram) {
  const uint32_t n = col->segments[chunk_id].size;
  const uint64_t radix_mask = ((uint64_t)1 << radix_bits) - 1;
  out->elements = (element_t*)malloc(sizeof(element_t) * (n ? n : 1));
  out->nulls = keep_nulls ? (uint8_t*)calloc(n ? n : 1, 1) : NULL;
  uint64_t count = 0;
  for (uint32_t i = 0; i < n; ++i) {
    int64_t value;
    int is_null;
    if (is_float_type(g_hashed_type)) {
      uint32_t column_type;
      const hy_row_id id = {chunk_id, i};
      const typed_value_t v = column_typed_value(col, id, &column_type);
      is_null = v.is_null;
      value = is_null ? 0 : key_of(v, column_type);
    } else {
      is_null = column_value(col, chunk_id, i, &value);
    }
    if (is_null && !keep_nulls) continue;
    const uint64_t hash = hyo_std_hash(value, g_hashed_type);
    if (!is_null && bloom_in && !bloom_get(bloom_in, hash) && !keep_nulls) continue; /* :354-358 */
    bloom_set(bloom_out, hash);                                                      /* :362 */
    out->elements[count].row_id.chunk_id = chunk_id;  /* reference segments: index in the segment (:364-371) */
    out->elements[count].row_id.chunk_offset = i;
    out->elements[count].value = value;
    if (keep_nulls) out->nulls[count] = (uint8_t)is_null;
    ++count;
    if (radix_bits > 0) ++histogram[hash & radix_mask];
  }
  out->size = count;
}

uint64_t hyo_join_materialize(const hyo_column* column, int keep_nulls, uint32_t radix_bits, const uint64_t* bloom_in,
                              uint64_t* bloom_out, hy_row_id* row_ids_out, int64_t* values_out, uint8_t* nulls_out,
                              uint64_t* chunk_element_counts_out, uint64_t* histograms_out) {
  const uint64_t partitions = (uint64_t)1 << radix_bits;
  uint64_t total = 0;
  g_hashed_type = HY_TYPE_LONG;   /* the step-level entry point is for integer columns */
  for (uint32_t c = 0; c < column->n_chunks; ++c) {
    partition_t p;
    uint64_t* hist = (uint64_t*)calloc(partitions, sizeof(uint64_t));
    materialize_chunk(column, c, keep_nulls, radix_bits, bloom_in, bloom_out, &p, hist);
    for (uint64_t i = 0; i < p.size; ++i) {
      row_ids_out[total + i] = p.elements[i].row_id;
      values_out[total + i] = p.elements[i].value;
      if (nulls_out) nulls_out[total + i] = p.nulls ? p.nulls[i] : 0;
    }
    if (chunk_element_counts_out) chunk_element_counts_out[c] = p.size;
    if (histograms_out) memcpy(histograms_out + (uint64_t)c * partitions, hist, sizeof(uint64_t) * partitions);
    total += p.size;
    free(hist); free(p.elements); free(p.nulls);
  }
  return total;
}

/* ---- tiny parallel-for (JobTask fan-out) ---------------------------------------------------------------------- */
typedef void (*job_fn)(void* ctx, uint64_t index);
typedef struct { job_fn fn; void* ctx; uint64_t n; volatile uint64_t* next; } pool_t;
static void* pool_worker(void* arg) {
  pool_t* p = (pool_t*)arg;
  for (;;) {
    const uint64_t i = __atomic_fetch_add(p->next, 1, __ATOMIC_RELAXED);
    if (i >= p->n) return NULL;
    p->fn(p->ctx, i);
  }
}
static void parallel_for(uint64_t n, job_fn fn, void* ctx, int threads) {
  if (threads <= 1 || n <= 1) { for (uint64_t i = 0; i < n; ++i) fn(ctx, i); return; }
  volatile uint64_t next = 0;
  pool_t pool = {fn, ctx, n, &next};
  if ((uint64_t)threads > n) threads = (int)n;
  pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (int t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, pool_worker, &pool);
  for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
  free(tids);
}

/* ---- PosHashTable (join_hash_steps.hpp:97-236): key -> dense id in insertion order; id -> RowIDs in insertion order */
typedef struct {
  int64_t* keys; uint32_t* ids; uint8_t* used; uint64_t capacity; /* open addressing */
  uint32_t distinct;
  uint64_t* offsets;   /* [distinct + 1] into pos_list (finalize()) */
  hy_row_id* pos_list;
  int exists;
} hash_table_t;

static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }

static uint32_t table_find(const hash_table_t* t, int64_t key) {
  if (!t->exists || t->capacity == 0 || key_is_nan(key)) return 0xFFFFFFFFu;
  uint64_t slot = mix((uint64_t)key) & (t->capacity - 1);
  while (t->used[slot]) {
    if (t->keys[slot] == key) return t->ids[slot];
    slot = (slot + 1) & (t->capacity - 1);
  }
  return 0xFFFFFFFFu;
}

static void table_build(hash_table_t* t, const partition_t* parts, uint32_t n_parts, const uint64_t* probe_bloom,
                        int all_positions) {
  uint64_t total = 0;
  for (uint32_t p = 0; p < n_parts; ++p) total += parts[p].size;
  uint64_t cap = 16;
  while (cap < total * 2) cap <<= 1;
  t->capacity = cap;
  t->keys = (int64_t*)malloc(sizeof(int64_t) * cap);
  t->ids = (uint32_t*)malloc(sizeof(uint32_t) * cap);
  t->used = (uint8_t*)calloc(cap, 1);
  t->distinct = 0;
  t->exists = 1;
  uint32_t* element_ids = (uint32_t*)malloc(sizeof(uint32_t) * (total ? total : 1));
  uint32_t* counts = (uint32_t*)calloc(total ? total : 1, sizeof(uint32_t));
  uint64_t e = 0;
  for (uint32_t p = 0; p < n_parts; ++p) {
    for (uint64_t i = 0; i < parts[p].size; ++i, ++e) {
      const int64_t key = parts[p].elements[i].value;
      element_ids[e] = 0xFFFFFFFFu;
      if (!bloom_get(probe_bloom, hyo_std_hash(key, g_hashed_type))) continue;   /* :476-479 */
      if (key_is_nan(key)) continue;   /* (the reference adds an entry that no lookup ever finds) */
      uint64_t slot = mix((uint64_t)key) & (cap - 1);
      while (t->used[slot] && t->keys[slot] != key) slot = (slot + 1) & (cap - 1);
      if (!t->used[slot]) { t->used[slot] = 1; t->keys[slot] = key; t->ids[slot] = t->distinct++; }
      element_ids[e] = t->ids[slot];
      counts[t->ids[slot]]++;
    }
  }
  if (all_positions) { /* finalize(): concatenate the per-id lists, each in insertion order (:147-175) */
    t->offsets = (uint64_t*)malloc(sizeof(uint64_t) * ((uint64_t)t->distinct + 1));
    uint64_t sum = 0;
    for (uint32_t id = 0; id < t->distinct; ++id) { t->offsets[id] = sum; sum += counts[id]; }
    t->offsets[t->distinct] = sum;
    t->pos_list = (hy_row_id*)malloc(sizeof(hy_row_id) * (sum ? sum : 1));
    uint64_t* cursor = (uint64_t*)malloc(sizeof(uint64_t) * ((uint64_t)t->distinct + 1));
    memcpy(cursor, t->offsets, sizeof(uint64_t) * ((uint64_t)t->distinct + 1));
    e = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
      for (uint64_t i = 0; i < parts[p].size; ++i, ++e) {
        if (element_ids[e] != 0xFFFFFFFFu) t->pos_list[cursor[element_ids[e]]++] = parts[p].elements[i].row_id;
      }
    }
    free(cursor);
  }
  free(element_ids); free(counts);
}

static void table_free(hash_table_t* t) {
  free(t->keys); free(t->ids); free(t->used); free(t->offsets); free(t->pos_list);
  memset(t, 0, sizeof(*t));
}

/* ---- pipeline -------------------------------------------------------------------------------------------------- */
typedef struct {
  const hyo_column* column; int keep_nulls; uint32_t radix_bits; const uint64_t* bloom_in;
  uint64_t** local_blooms; partition_t* chunks; uint64_t* histograms; uint64_t partitions;
} materialize_ctx_t;

static void materialize_job(void* arg, uint64_t c) {
  materialize_ctx_t* m = (materialize_ctx_t*)arg;
  m->local_blooms[c] = (uint64_t*)calloc(BLOOM_WORDS, sizeof(uint64_t));
  materialize_chunk(m->column, (uint32_t)c, m->keep_nulls, m->radix_bits, m->bloom_in, m->local_blooms[c], &m->chunks[c],
                    m->histograms + c * m->partitions);
}

static partition_t* materialize(const hyo_column* col, int keep_nulls, uint32_t radix_bits, const uint64_t* bloom_in,
                                uint64_t* bloom_out, uint64_t** histograms_out, int threads) {
  const uint64_t partitions = (uint64_t)1 << radix_bits;
  partition_t* chunks = (partition_t*)calloc(col->n_chunks ? col->n_chunks : 1, sizeof(partition_t));
  uint64_t* hist = (uint64_t*)calloc((uint64_t)(col->n_chunks ? col->n_chunks : 1) * partitions, sizeof(uint64_t));
  uint64_t** local = (uint64_t**)calloc(col->n_chunks ? col->n_chunks : 1, sizeof(uint64_t*));
  materialize_ctx_t ctx = {col, keep_nulls, radix_bits, bloom_in, local, chunks, hist, partitions};
  parallel_for(col->n_chunks, materialize_job, &ctx, threads);
  for (uint32_t c = 0; c < col->n_chunks; ++c) { /* output_bloom_filter |= local (:405-409) */
    for (uint32_t w = 0; w < BLOOM_WORDS; ++w) bloom_out[w] |= local[c][w];
    free(local[c]);
  }
  free(local);
  *histograms_out = hist;
  return chunks;
}

/* partition_by_radix (:509-617): stable scatter, offsets = prefix over input partitions per radix value. */
static partition_t* partition_by_radix(const partition_t* in, uint32_t n_in, const uint64_t* histograms,
                                       uint32_t radix_bits, int keep_nulls) {
  const uint64_t partitions = (uint64_t)1 << radix_bits;
  const uint64_t mask = partitions - 1;
  partition_t* out = (partition_t*)calloc(partitions, sizeof(partition_t));
  uint64_t* offsets = (uint64_t*)malloc(sizeof(uint64_t) * (uint64_t)(n_in ? n_in : 1) * partitions);
  for (uint64_t r = 0; r < partitions; ++r) {
    uint64_t size = 0;
    for (uint32_t c = 0; c < n_in; ++c) { offsets[(uint64_t)c * partitions + r] = size; size += histograms[(uint64_t)c * partitions + r]; }
    out[r].elements = (element_t*)malloc(sizeof(element_t) * (size ? size : 1));
    out[r].nulls = keep_nulls ? (uint8_t*)malloc(size ? size : 1) : NULL;
    out[r].size = size;
  }
  for (uint32_t c = 0; c < n_in; ++c) {
    for (uint64_t i = 0; i < in[c].size; ++i) {
      const uint64_t r = hyo_std_hash(in[c].elements[i].value, g_hashed_type) & mask;
      const uint64_t idx = offsets[(uint64_t)c * partitions + r]++;
      out[r].elements[idx] = in[c].elements[i];
      if (keep_nulls) out[r].nulls[idx] = in[c].nulls[i];
    }
  }
  free(offsets);
  return out;
}

static void free_partitions(partition_t* p, uint64_t n) {
  if (!p) return;
  for (uint64_t i = 0; i < n; ++i) { free(p[i].elements); free(p[i].nulls); }
  free(p);
}

typedef struct { hy_row_id* build; hy_row_id* probe; uint64_t size, capacity; } pos_pair_t;
static void pair_push(pos_pair_t* p, hy_row_id b, hy_row_id r, int with_build) {
  if (p->size == p->capacity) {
    p->capacity = p->capacity ? p->capacity * 2 : 64;
    p->probe = (hy_row_id*)realloc(p->probe, sizeof(hy_row_id) * p->capacity);
    if (with_build) p->build = (hy_row_id*)realloc(p->build, sizeof(hy_row_id) * p->capacity);
  }
  p->probe[p->size] = r;
  if (with_build) p->build[p->size] = b;
  p->size++;
}

typedef struct {
  const partition_t* probe_partitions; const hash_table_t* tables; uint32_t n_tables; uint32_t mode; int keep_nulls;
  uint64_t build_rows;
  struct { uint32_t partition; uint64_t begin, end; } * slices;
  pos_pair_t* out;
  const secondary_t* secondary; uint32_t n_secondary;
} probe_ctx_t;

static const hy_row_id NULL_ROW = {0xFFFFFFFFu, 0xFFFFFFFFu};

static void probe_job(void* arg, uint64_t s) {
  probe_ctx_t* p = (probe_ctx_t*)arg;
  const partition_t* part = &p->probe_partitions[p->slices[s].partition];
  const hash_table_t* table = NULL;
  if (p->n_tables) {
    const uint32_t idx = p->n_tables > 1 ? p->slices[s].partition : 0;
    if (p->tables[idx].exists) table = &p->tables[idx];
  }
  pos_pair_t* out = &p->out[s];
  const uint32_t mode = p->mode;
  const int semi_anti = mode == HY_JOIN_SEMI || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE;
  for (uint64_t i = p->slices[s].begin; i < p->slices[s].end; ++i) {
    const element_t* e = &part->elements[i];
    const int is_null = p->keep_nulls && part->nulls[i];
    if (!semi_anti) { /* probe() :690-777 */
      if (!table) { if (p->keep_nulls) pair_push(out, NULL_ROW, e->row_id, 1); continue; }
      if (mode == HY_JOIN_INNER && e->row_id.chunk_offset == 0xFFFFFFFFu) continue;
      const uint32_t id = table_find(table, e->value);
      if (id != 0xFFFFFFFFu) {
        if (is_null) { pair_push(out, NULL_ROW, e->row_id, 1); continue; }
        if (!p->n_secondary) {
          for (uint64_t m = table->offsets[id]; m < table->offsets[id + 1]; ++m) pair_push(out, table->pos_list[m], e->row_id, 1);
        } else { /* :727-747 */
          int match_found = 0;
          for (uint64_t m = table->offsets[id]; m < table->offsets[id + 1]; ++m) {
            if (!satisfies_all_predicates(p->secondary, p->n_secondary, table->pos_list[m], e->row_id)) continue;
            pair_push(out, table->pos_list[m], e->row_id, 1);
            match_found = 1;
          }
          if (p->keep_nulls && !match_found) pair_push(out, NULL_ROW, e->row_id, 1);
        }
      } else if (p->keep_nulls) {
        pair_push(out, NULL_ROW, e->row_id, 1);
      }
    } else { /* probe_semi_anti() :844-908 */
      if (!table) {
        if (mode == HY_JOIN_ANTI_NULL_AS_FALSE) pair_push(out, NULL_ROW, e->row_id, 0);
        else if (mode == HY_JOIN_ANTI_NULL_AS_TRUE && !(is_null && p->build_rows != 0)) pair_push(out, NULL_ROW, e->row_id, 0);
        continue;
      }
      if (mode == HY_JOIN_SEMI) { if (e->row_id.chunk_offset == 0xFFFFFFFFu) continue; }
      else if (mode == HY_JOIN_ANTI_NULL_AS_FALSE) { if (is_null) { pair_push(out, NULL_ROW, e->row_id, 0); continue; } }
      else if (is_null) continue;
      int matches;
      if (!p->n_secondary) {
        matches = table_find(table, e->value) != 0xFFFFFFFFu;
      } else { /* :869-876 */
        matches = 0;
        const uint32_t id = table_find(table, e->value);
        if (id != 0xFFFFFFFFu) {
          for (uint64_t m = table->offsets[id]; m < table->offsets[id + 1] && !matches; ++m)
            matches = satisfies_all_predicates(p->secondary, p->n_secondary, table->pos_list[m], e->row_id);
        }
      }
      if ((mode == HY_JOIN_SEMI && matches) || (mode != HY_JOIN_SEMI && !matches)) pair_push(out, NULL_ROW, e->row_id, 0);
    }
  }
}

typedef struct { hash_table_t* tables; const partition_t* build_partitions; const uint64_t* probe_bloom; int all_positions; } build_ctx_t;
static void build_job(void* arg, uint64_t r) {
  build_ctx_t* b = (build_ctx_t*)arg;
  if (b->build_partitions[r].size == 0) return; /* :458-460 */
  table_build(&b->tables[r], &b->build_partitions[r], 1, b->probe_bloom, b->all_positions);
}

int32_t hyo_join_hash(const hyo_column* left, const hyo_column* right, uint32_t mode, hy_join_result* result,
                      int threads) {
  return hyo_join_hash_predicates(left, right, mode, NULL, 0, result, threads);
}

int32_t hyo_join_hash_predicates(const hyo_column* left, const hyo_column* right, uint32_t mode, const hy_join_predicate* secondary_in,
                                 uint32_t n_secondary, hy_join_result* result, int threads) {
  if (mode == HY_JOIN_FULL_OUTER || mode == HY_JOIN_CROSS || mode > HY_JOIN_ANTI_NULL_AS_FALSE) return HY_ERR_UNSUPPORTED;
  if (n_secondary > HY_MAX_SECONDARY_PREDICATES || (n_secondary && mode == HY_JOIN_ANTI_NULL_AS_TRUE)) return HY_ERR_UNSUPPORTED;   /* join_hash.cpp:39-44 */
  {
    const uint32_t left_type = left->n_chunks ? left->segments[0].data_type : HY_TYPE_LONG;
    const uint32_t right_type = right->n_chunks ? right->segments[0].data_type : left_type;
    if (left_type < HY_TYPE_INT || left_type > HY_TYPE_DOUBLE || right_type < HY_TYPE_INT || right_type > HY_TYPE_DOUBLE) return HY_ERR_UNSUPPORTED;
    g_hashed_type = hashed_type_of(left->n_chunks ? left_type : right_type, right_type);
  }
  uint64_t left_rows = 0, right_rows = 0;
  for (uint32_t c = 0; c < left->n_chunks; ++c) left_rows += left->segments[c].size;
  for (uint32_t c = 0; c < right->n_chunks; ++c) right_rows += right->segments[c].size;
  /* join_hash.cpp:139-155 */
  const int build_right = mode == HY_JOIN_LEFT || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE ||
                          mode == HY_JOIN_SEMI || (mode == HY_JOIN_INNER && left_rows > right_rows);
  const hyo_column* build = build_right ? right : left;
  const hyo_column* probe = build_right ? left : right;
  const uint64_t build_rows = build_right ? right_rows : left_rows, probe_rows = build_right ? left_rows : right_rows;
  uint32_t radix_bits = result->radix_bits == 0xFFFFFFFFu ? hyo_calculate_radix_bits(build_rows, probe_rows) : result->radix_bits;
  result->radix_bits = radix_bits;
  result->left_is_build = (uint32_t)!build_right;
  const uint64_t partitions = (uint64_t)1 << radix_bits;
  const int keep_nulls_build = mode == HY_JOIN_ANTI_NULL_AS_TRUE;                          /* :284-286 */
  const int keep_nulls_probe = mode == HY_JOIN_LEFT || mode == HY_JOIN_RIGHT || mode == HY_JOIN_ANTI_NULL_AS_TRUE ||
                               mode == HY_JOIN_ANTI_NULL_AS_FALSE;
  const int semi_anti = mode == HY_JOIN_SEMI || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE;

  uint64_t* build_bloom = (uint64_t*)calloc(BLOOM_WORDS, sizeof(uint64_t));
  uint64_t* probe_bloom = (uint64_t*)calloc(BLOOM_WORDS, sizeof(uint64_t));
  uint64_t *hist_build = NULL, *hist_probe = NULL;
  partition_t *mat_build, *mat_probe;
  if (build_rows < probe_rows) { /* :365-381 */
    mat_build = materialize(build, keep_nulls_build, radix_bits, NULL, build_bloom, &hist_build, threads);
    mat_probe = materialize(probe, keep_nulls_probe, radix_bits, build_bloom, probe_bloom, &hist_probe, threads);
  } else {
    mat_probe = materialize(probe, keep_nulls_probe, radix_bits, NULL, probe_bloom, &hist_probe, threads);
    mat_build = materialize(build, keep_nulls_build, radix_bits, probe_bloom, build_bloom, &hist_build, threads);
  }
  partition_t *radix_build, *radix_probe;
  uint64_t n_build_parts, n_probe_parts;
  if (radix_bits > 0) { /* :398-435 */
    radix_build = build->n_chunks ? partition_by_radix(mat_build, build->n_chunks, hist_build, radix_bits, keep_nulls_build) : NULL;
    radix_probe = probe->n_chunks ? partition_by_radix(mat_probe, probe->n_chunks, hist_probe, radix_bits, keep_nulls_probe) : NULL;
    n_build_parts = build->n_chunks ? partitions : 0;   /* empty radix_container stays empty (:512-514) */
    n_probe_parts = probe->n_chunks ? partitions : 0;
    free_partitions(mat_build, build->n_chunks ? build->n_chunks : 1);
    free_partitions(mat_probe, probe->n_chunks ? probe->n_chunks : 1);
  } else {
    radix_build = mat_build; radix_probe = mat_probe;
    n_build_parts = build->n_chunks; n_probe_parts = probe->n_chunks;
  }
  free(hist_build); free(hist_probe);

  /* secondary predicates follow the side swap (join_hash.cpp:158-165) */
  secondary_t secondary[HY_MAX_SECONDARY_PREDICATES];
  for (uint32_t p = 0; p < n_secondary; ++p) {
    const hyo_column* l = (const hyo_column*)secondary_in[p].left_column;
    const hyo_column* r = (const hyo_column*)secondary_in[p].right_column;
    if (secondary_in[p].condition > HY_PRED_GREATER_THAN_EQUALS) return HY_ERR_INVALID;
    secondary[p].build = build_right ? r : l;
    secondary[p].probe = build_right ? l : r;
    secondary[p].condition = build_right ? flip_condition(secondary_in[p].condition) : secondary_in[p].condition;
  }

  /* build (:426-507); semi / anti joins with secondary predicates need every position (join_hash.cpp:445-453) */
  const int all_positions = !semi_anti || n_secondary != 0;
  uint32_t n_tables = 0;
  hash_table_t* tables = NULL;
  if (n_build_parts > 0) {
    if (radix_bits == 0) {
      n_tables = 1;
      tables = (hash_table_t*)calloc(1, sizeof(hash_table_t));
      table_build(&tables[0], radix_build, (uint32_t)n_build_parts, probe_bloom, all_positions);
    } else {
      n_tables = (uint32_t)n_build_parts;
      tables = (hash_table_t*)calloc(n_tables, sizeof(hash_table_t));
      build_ctx_t bctx = {tables, radix_build, probe_bloom, all_positions};
      parallel_for(n_tables, build_job, &bctx, threads);
    }
  }

  int32_t status = HY_OK;
  result->n_slices = 0;
  result->n_pairs = 0;
  int early_out = 0;
  if (mode == HY_JOIN_ANTI_NULL_AS_TRUE) { /* :483-494 */
    for (uint64_t p = 0; p < n_build_parts && !early_out; ++p)
      for (uint64_t i = 0; i < radix_build[p].size; ++i) if (radix_build[p].nulls[i]) { early_out = 1; break; }
  }
  if (!early_out) {
    /* probe (:641-660): one slice per PROBE_SIZE_PER_CHUNK elements of every non-empty partition */
    uint64_t n_slices = 0;
    for (uint64_t p = 0; p < n_probe_parts; ++p) n_slices += (radix_probe[p].size + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK;
    probe_ctx_t pctx;
    memset(&pctx, 0, sizeof(pctx));
    pctx.probe_partitions = radix_probe; pctx.tables = tables; pctx.n_tables = n_tables; pctx.mode = mode;
    pctx.keep_nulls = keep_nulls_probe; pctx.build_rows = build_rows;
    pctx.secondary = secondary; pctx.n_secondary = n_secondary;
    pctx.slices = malloc(sizeof(*pctx.slices) * (n_slices ? n_slices : 1));
    pctx.out = (pos_pair_t*)calloc(n_slices ? n_slices : 1, sizeof(pos_pair_t));
    uint64_t s = 0;
    for (uint64_t p = 0; p < n_probe_parts; ++p)
      for (uint64_t b = 0; b < radix_probe[p].size; b += PROBE_SIZE_PER_CHUNK, ++s) {
        pctx.slices[s].partition = (uint32_t)p;
        pctx.slices[s].begin = b;
        pctx.slices[s].end = b + PROBE_SIZE_PER_CHUNK < radix_probe[p].size ? b + PROBE_SIZE_PER_CHUNK : radix_probe[p].size;
      }
    parallel_for(n_slices, probe_job, &pctx, threads);
    uint64_t total = 0;
    for (s = 0; s < n_slices; ++s) total += pctx.out[s].size;
    if (n_slices > result->slice_capacity || total > result->capacity) status = HY_ERR_CAPACITY;
    else {
      uint64_t cursor = 0;
      for (s = 0; s < n_slices; ++s) {
        result->slice_offsets[s] = cursor;
        if (pctx.out[s].size) {
          hy_row_id* build_out = result->left_is_build ? result->left_pos : result->right_pos;
          hy_row_id* probe_out = result->left_is_build ? result->right_pos : result->left_pos;
          memcpy(probe_out + cursor, pctx.out[s].probe, sizeof(hy_row_id) * pctx.out[s].size);
          if (!semi_anti && build_out) memcpy(build_out + cursor, pctx.out[s].build, sizeof(hy_row_id) * pctx.out[s].size);
        }
        cursor += pctx.out[s].size;
      }
      result->slice_offsets[n_slices] = cursor;
      result->n_slices = (uint32_t)n_slices;
      result->n_pairs = total;
    }
    for (s = 0; s < n_slices; ++s) { free(pctx.out[s].build); free(pctx.out[s].probe); }
    free(pctx.out); free(pctx.slices);
  } else {
    result->slice_offsets[0] = 0;
  }
  for (uint32_t t = 0; t < n_tables; ++t) table_free(&tables[t]);
  free(tables);
  free_partitions(radix_build, n_build_parts ? n_build_parts : 1);
  free_partitions(radix_probe, n_probe_parts ? n_probe_parts : 1);
  free(build_bloom); free(probe_bloom);
  return status;
}

/* ---- output chunking (join_output_writing.cpp:205-340) --------------------------------------------------------------
 * write_output_chunks walks the per-partition PosLists (here: the slices of hy_join_result.slice_offsets) in order and emits
 * one output chunk per non-empty PosList -- after merging, when allow_partition_merge is set (JoinHash always sets it,
 * join_hash.cpp:563), a PosList smaller than MIN_SIZE = 1000 with its successors as long as the sum stays below
 * MAX_SIZE = 4000 (:245-296).  chunk_offsets[k] .. chunk_offsets[k + 1] are the pairs of output chunk k; returns the number
 * of chunks (chunk_offsets needs room for n_slices + 1 values). */
uint32_t hyo_write_output_chunks(const uint64_t* slice_offsets, uint32_t n_slices, int allow_partition_merge, uint64_t* chunk_offsets) {
  const uint64_t MIN_SIZE = 1000, MAX_SIZE = MIN_SIZE * 4;
  uint32_t partition_id = 0, chunk_input_position = 0;
  chunk_offsets[0] = 0;
  while (partition_id < n_slices) {
    uint64_t size = slice_offsets[partition_id + 1] - slice_offsets[partition_id];   /* right_side_pos_list->size() */
    if (size == 0) {                                    /* :270-273 */
      ++partition_id;
      continue;
    }
    if (allow_partition_merge) {
      while (partition_id + 1 < n_slices && size < MIN_SIZE &&
             size + (slice_offsets[partition_id + 2] - slice_offsets[partition_id + 1]) < MAX_SIZE) {   /* :278-279 */
        size += slice_offsets[partition_id + 2] - slice_offsets[partition_id + 1];
        ++partition_id;
      }
    }
    chunk_offsets[chunk_input_position + 1] = slice_offsets[partition_id + 1];
    ++partition_id;
    ++chunk_input_position;
  }
  return chunk_input_position;
}
