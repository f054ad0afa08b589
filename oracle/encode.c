/*
 * encode.c -- restatement of the segment encoders, so that test inputs are laid out exactly as Hyrise lays them out.
 * TEST INFRASTRUCTURE ONLY (see hy_oracle.h).
 */
#include <stdlib.h>
#include <string.h>

#include "hy_oracle.h"

#define DEFINE_CMP(NAME, T)                       \
  static int NAME(const void* a, const void* b) { \
    const T x = *(const T*)a, y = *(const T*)b;   \
    return (x < y) ? -1 : (x > y) ? 1 : 0;        \
  }
DEFINE_CMP(cmp_i32, int32_t)
DEFINE_CMP(cmp_i64, int64_t)
DEFINE_CMP(cmp_f32, float)
DEFINE_CMP(cmp_f64, double)

static size_t type_size(uint32_t data_type) {
  switch (data_type) {
    case HY_TYPE_INT: return 4;
    case HY_TYPE_LONG: return 8;
    case HY_TYPE_FLOAT: return 4;
    case HY_TYPE_DOUBLE: return 8;
    default: return 0;
  }
}

/* FixedWidthIntegerCompressor::_compress_using_max_value (fixed_width_integer_compressor.cpp:33-44). */
static uint32_t width_for_max(uint32_t max_value) {
  if (max_value <= 0xFFu) return 1;
  if (max_value <= 0xFFFFu) return 2;
  return 4;
}

static void store_compressed(void* out, uint32_t width, uint32_t idx, uint32_t v) {
  if (width == 1) ((uint8_t*)out)[idx] = (uint8_t)v;
  else if (width == 2) ((uint16_t*)out)[idx] = (uint16_t)v;
  else ((uint32_t*)out)[idx] = v;
}

#define LOWER_BOUND(T, dict, d, value, result)         \
  do {                                                 \
    uint32_t lo_ = 0, hi_ = (d);                       \
    while (lo_ < hi_) {                                \
      const uint32_t mid_ = lo_ + (hi_ - lo_) / 2;     \
      if (((const T*)(dict))[mid_] < (value)) lo_ = mid_ + 1; \
      else hi_ = mid_;                                 \
    }                                                  \
    (result) = lo_;                                    \
  } while (0)

/* DictionaryEncoder::on_encode (dictionary_encoder.hpp:33-103): dense values -> sort -> unique; value id via
 * lower_bound; NULL = dictionary.size(); width from max value id == null value id (:85-92). */
uint32_t hyo_encode_dictionary(uint32_t data_type, const void* values, const uint8_t* nulls, uint32_t n,
                               void* dict_out, void* av_out, uint32_t* width_out) {
  const size_t ts = type_size(data_type);
  if (ts == 0) return 0xFFFFFFFFu;
  uint32_t dense = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (nulls && nulls[i]) continue;
    memcpy((char*)dict_out + (size_t)dense * ts, (const char*)values + (size_t)i * ts, ts);
    ++dense;
  }
  switch (data_type) {
    case HY_TYPE_INT: qsort(dict_out, dense, ts, cmp_i32); break;
    case HY_TYPE_LONG: qsort(dict_out, dense, ts, cmp_i64); break;
    case HY_TYPE_FLOAT: qsort(dict_out, dense, ts, cmp_f32); break;
    default: qsort(dict_out, dense, ts, cmp_f64); break;
  }
  uint32_t d = 0;
  for (uint32_t i = 0; i < dense; ++i) { /* std::unique (operator==) */
    int equal_prev = 0;
    if (d > 0) {
      switch (data_type) {
        case HY_TYPE_INT: equal_prev = ((int32_t*)dict_out)[d - 1] == ((int32_t*)dict_out)[i]; break;
        case HY_TYPE_LONG: equal_prev = ((int64_t*)dict_out)[d - 1] == ((int64_t*)dict_out)[i]; break;
        case HY_TYPE_FLOAT: equal_prev = ((float*)dict_out)[d - 1] == ((float*)dict_out)[i]; break;
        default: equal_prev = ((double*)dict_out)[d - 1] == ((double*)dict_out)[i]; break;
      }
    }
    if (!equal_prev) {
      memmove((char*)dict_out + (size_t)d * ts, (char*)dict_out + (size_t)i * ts, ts);
      ++d;
    }
  }
  const uint32_t null_value_id = d;
  const uint32_t width = width_for_max(null_value_id);
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t vid = null_value_id;
    if (!(nulls && nulls[i])) {
      switch (data_type) {
        case HY_TYPE_INT: LOWER_BOUND(int32_t, dict_out, d, ((const int32_t*)values)[i], vid); break;
        case HY_TYPE_LONG: LOWER_BOUND(int64_t, dict_out, d, ((const int64_t*)values)[i], vid); break;
        case HY_TYPE_FLOAT: LOWER_BOUND(float, dict_out, d, ((const float*)values)[i], vid); break;
        default: LOWER_BOUND(double, dict_out, d, ((const double*)values)[i], vid); break;
      }
    }
    store_compressed(av_out, width, i, vid);
  }
  *width_out = width;
  return d;
}

/* FrameOfReferenceEncoder::on_encode (frame_of_reference_encoder.hpp:25-122): blocks of 2048, minimum over non-NULL
 * values of the block (INT32_MAX if the block holds only NULLs), NULL rows store offset 0 (value := minimum :96-103),
 * width from the maximum offset, null vector kept only if any NULL was seen (:113-120). */
uint32_t hyo_encode_frame_of_reference(const int32_t* values, const uint8_t* nulls, uint32_t n, int32_t* minima_out,
                                       void* offsets_out, uint32_t* has_nulls_out) {
  uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t max_offset = 0;
  int any_null = 0;
  for (uint32_t begin = 0, block = 0; begin < n; begin += HY_FOR_BLOCK_SIZE, ++block) {
    const uint32_t end = (begin + HY_FOR_BLOCK_SIZE < n) ? begin + HY_FOR_BLOCK_SIZE : n;
    int32_t min_value = INT32_MAX;
    for (uint32_t i = begin; i < end; ++i) {
      const int is_null = nulls && nulls[i];
      any_null |= is_null;
      if (!is_null && values[i] < min_value) min_value = values[i];
    }
    minima_out[block] = min_value;
    for (uint32_t i = begin; i < end; ++i) {
      const int is_null = nulls && nulls[i];
      const int32_t value = is_null ? min_value : values[i];
      const uint32_t offset = (uint32_t)value - (uint32_t)min_value;
      tmp[i] = offset;
      if (offset > max_offset) max_offset = offset;
    }
  }
  const uint32_t width = width_for_max(max_offset);
  for (uint32_t i = 0; i < n; ++i) store_compressed(offsets_out, width, i, tmp[i]);
  free(tmp);
  *has_nulls_out = (uint32_t)any_null;
  return width;
}

void hyo_pack_nulls(const uint8_t* nulls, uint32_t n, uint64_t* words_out) {
  const uint32_t words = (n + 63) / 64;
  for (uint32_t w = 0; w < words; ++w) words_out[w] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (nulls[i]) words_out[i / 64] |= (uint64_t)1 << (i % 64);
  }
}
