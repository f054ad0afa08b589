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

}  // namespace hy
