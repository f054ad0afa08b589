// hy_decode.hpp -- row-at-a-time decoding of a column cell on the device (generic paths of AggregateHash and Projection;
// the hot paths have their own batched / streaming decoders).
#pragma once
#include "hy_device.hpp"

namespace hy {

__device__ __forceinline__ uint32_t aload_compressed(const void* data, uint32_t width, uint32_t i) {
  if (width == 1) return static_cast<const uint8_t*>(data)[i];
  if (width == 2) return static_cast<const uint16_t*>(data)[i];
  return static_cast<const uint32_t*>(data)[i];
}

struct Value {
  bool is_null;
  int64_t i;
  double f;
};

__device__ inline Value data_value(const DevSegment& s, uint32_t row) {
  Value v{false, 0, 0.0};
  const void* values = s.data;
  uint32_t index = row;
  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = aload_compressed(s.data, s.width, row);
    if (vid >= s.aux_size) { v.is_null = true; return v; }
    values = s.aux;
    index = vid;
  } else {
    if (s.nulls && ((s.nulls[row >> 6] >> (row & 63)) & 1)) { v.is_null = true; return v; }
    if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
      v.i = static_cast<int32_t>(aload_compressed(s.data, s.width, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]));
      return v;
    }
  }
  switch (s.data_type) {
    case HY_TYPE_INT: v.i = static_cast<const int32_t*>(values)[index]; break;
    case HY_TYPE_LONG: v.i = static_cast<const int64_t*>(values)[index]; break;
    case HY_TYPE_FLOAT: v.f = static_cast<const float*>(values)[index]; break;
    default: v.f = static_cast<const double*>(values)[index]; break;
  }
  return v;
}

__device__ inline Value column_value(const DevSegment* segments, uint32_t chunk, uint32_t row) {
  const DevSegment& s = segments[chunk];
  if (s.encoding != HY_ENC_REFERENCE) return data_value(s, row);
  hy_row_id r;
  if (s.data) r = static_cast<const hy_row_id*>(s.data)[row];
  else { r.chunk_id = s.ref_chunk_id; r.chunk_offset = row; }
  if (r.chunk_offset == 0xFFFFFFFFu) return Value{true, 0, 0.0};
  return data_value(s.ref[r.chunk_id], r.chunk_offset);
}

// ---- ReferenceSegments whose PosList references ONE chunk (what scans, Validate and join output chunks guarantee) ------------
// Resolved once per slice by the caller: the REFERENCED segment's descriptor (a uniform load: scalar registers) and the PosList,
// whose offsets then replace the batch's row numbers (dereference_rows) before the batched decoder below runs on the referenced
// segment.  *pos_words stays nullptr for data segments, for EntireChunkPosLists (rows map one to one) and for PosLists over
// several chunks (those keep the segment as it is: the row-by-row path of decode_rows).
__device__ __forceinline__ DevSegment resolve_segment(const DevSegment& s, const uint32_t** pos_words) {
  *pos_words = nullptr;
  if (s.encoding != HY_ENC_REFERENCE || s.ref_chunk_id == 0xFFFFFFFFu) return s;
  *pos_words = static_cast<const uint32_t*>(s.data);
  return s.ref[s.ref_chunk_id];
}

// row[i] <- chunk offset of RowID row[i] of the PosList; returns the mask of NULL RowIDs (their row becomes 0).
template <int B>
__device__ __forceinline__ uint32_t dereference_rows(const uint32_t* pos_words, uint32_t (&row)[B]) {
  uint32_t null_rows = 0;
#pragma unroll
  for (int i = 0; i < B; ++i) row[i] = pos_words[2 * size_t{row[i]} + 1];
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (row[i] == 0xFFFFFFFFu) { null_rows |= 1u << i; row[i] = 0; }
  }
  return null_rows;
}

// ---- batched decoding: B rows of one column per call, the loads of all rows issued before any is used ----------------
// bits[i]: the value as int64 (integer columns) or as the bits of a double (float/double columns); null bit i set for NULL.
// `s` = segments[chunk], loaded by the caller (once for all the calls on one chunk: the descriptor is a dependent load in
// front of the data loads).
template <int B>
__device__ __forceinline__ void decode_rows(const DevSegment& s, const DevSegment* segments, uint32_t chunk, const uint32_t (&row)[B], uint32_t valid, uint64_t (&bits)[B],
                                            uint32_t* nulls) {
  *nulls = 0;
  if (s.encoding == HY_ENC_REFERENCE) {
#pragma unroll 1
    for (int i = 0; i < B; ++i) {
      bits[i] = 0;
      if (!((valid >> i) & 1)) continue;
      const Value v = column_value(segments, chunk, row[i]);
      const bool is_float = s.data_type == HY_TYPE_FLOAT || s.data_type == HY_TYPE_DOUBLE;
      if (v.is_null) *nulls |= 1u << i;
      else bits[i] = is_float ? static_cast<uint64_t>(__double_as_longlong(v.f)) : static_cast<uint64_t>(v.i);
    }
    return;
  }
  const void* values = s.data;
  uint32_t index[B];
#pragma unroll
  for (int i = 0; i < B; ++i) index[i] = row[i];
  if (s.encoding == HY_ENC_DICTIONARY) {
    if (s.width == 1) {
#pragma unroll
      for (int i = 0; i < B; ++i) index[i] = static_cast<const uint8_t*>(s.data)[row[i]];
    } else if (s.width == 2) {
#pragma unroll
      for (int i = 0; i < B; ++i) index[i] = static_cast<const uint16_t*>(s.data)[row[i]];
    } else {
#pragma unroll
      for (int i = 0; i < B; ++i) index[i] = static_cast<const uint32_t*>(s.data)[row[i]];
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      if (index[i] >= s.aux_size) { *nulls |= 1u << i; index[i] = 0; }
    }
    values = s.aux;
    if (s.aux_size == 0) {
#pragma unroll
      for (int i = 0; i < B; ++i) bits[i] = 0;
      return;
    }
  } else if (s.nulls) {
    uint64_t word[B];
#pragma unroll
    for (int i = 0; i < B; ++i) word[i] = s.nulls[row[i] >> 6];
#pragma unroll
    for (int i = 0; i < B; ++i) *nulls |= static_cast<uint32_t>((word[i] >> (row[i] & 63)) & 1) << i;
  }
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    uint32_t raw[B];
    int32_t bias[B];
    if (s.width == 1) {
#pragma unroll
      for (int i = 0; i < B; ++i) raw[i] = static_cast<const uint8_t*>(s.data)[row[i]];
    } else if (s.width == 2) {
#pragma unroll
      for (int i = 0; i < B; ++i) raw[i] = static_cast<const uint16_t*>(s.data)[row[i]];
    } else {
#pragma unroll
      for (int i = 0; i < B; ++i) raw[i] = static_cast<const uint32_t*>(s.data)[row[i]];
    }
#pragma unroll
    for (int i = 0; i < B; ++i) bias[i] = static_cast<const int32_t*>(s.aux)[row[i] / HY_FOR_BLOCK_SIZE];
#pragma unroll
    for (int i = 0; i < B; ++i) bits[i] = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(raw[i] + static_cast<uint32_t>(bias[i]))));
    return;
  }
  switch (s.data_type) {
    case HY_TYPE_INT: {
      int32_t v[B];
#pragma unroll
      for (int i = 0; i < B; ++i) v[i] = static_cast<const int32_t*>(values)[index[i]];
#pragma unroll
      for (int i = 0; i < B; ++i) bits[i] = static_cast<uint64_t>(static_cast<int64_t>(v[i]));
      break;
    }
    case HY_TYPE_LONG:
#pragma unroll
      for (int i = 0; i < B; ++i) bits[i] = static_cast<const uint64_t*>(values)[index[i]];
      break;
    case HY_TYPE_FLOAT: {
      float v[B];
#pragma unroll
      for (int i = 0; i < B; ++i) v[i] = static_cast<const float*>(values)[index[i]];
#pragma unroll
      for (int i = 0; i < B; ++i) bits[i] = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(v[i])));
      break;
    }
    default:
#pragma unroll
      for (int i = 0; i < B; ++i) bits[i] = static_cast<const uint64_t*>(values)[index[i]];
      break;
  }
}

template <int B>
__device__ __forceinline__ void decode_rows(const DevSegment* segments, uint32_t chunk, const uint32_t (&row)[B], uint32_t valid, uint64_t (&bits)[B], uint32_t* nulls) {
  const DevSegment s = segments[chunk];
  decode_rows<B>(s, segments, chunk, row, valid, bits, nulls);
}

}  // namespace hy
