// exchange.hip -- what the multi-GPU execution needs from one GPU: the reference is a single
// process; sharding its operators over GPUs adds one exchange step per operator, and the data of that step is produced and
// consumed here, on the device, so that the collective (RCCL over xGMI, driven by the host plumbing in
// hyrise_amd/distributed.py) moves device buffers and nothing crosses PCIe.
//
//   hy_column_export      a column's values decoded into one flat device array (+ NULL bytes): the build side of a
//                         broadcast-build join before its all-gather; any operator result handed to another library
//   hy_repartition_count  rows per destination of a hash repartition:  destination = std::hash(key) % parts  (std::hash of an
//   hy_repartition_pack   integer is the integer: JoinHash's own hash, join_hash_steps.hpp:352); then the (key, RowID) tuples
//                         grouped by destination, stable in (chunk, row) order -- the send buffer of the all-to-all
//   hy_gather_row_ids     positions in a received tuple array -> the RowIDs that travelled with the keys (the local join of a
//                         repartitioned join speaks positions of the received arrays)
#include "hy_device.hpp"
#include "hy_decode.hpp"

#include <algorithm>

namespace hy {

struct ExportArgs {
  const DevSegment* segments;
  const Slice* slices;
  const uint64_t* row_base;
  void* values;        // [rows] of the column's type
  uint8_t* nulls;      // [rows] or nullptr
  uint32_t width;      // bytes per value
  uint32_t is_float;   // the column holds float / double (the decoder returns them as doubles)
};

__global__ __launch_bounds__(256) void export_rows(ExportArgs a) {
  // (a chunk's eight slices on one XCD, so that one L2 fetches the chunk's dictionary: projection_rows' mapping)
  uint32_t slice_index = blockIdx.x;
  if ((blockIdx.x | 63u) < gridDim.x) slice_index = (blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u);
  const Slice slice = a.slices[slice_index];
  if (slice.row_count == 0) return;
  const DevSegment s = a.segments[slice.chunk];
  const uint64_t base = a.row_base[slice.chunk] + slice.row_begin;
  constexpr int BATCH = 8;
#pragma unroll 1
  for (uint32_t block = 0; block < SLICE_ROWS / 256 / BATCH; ++block) {
    if (block * BATCH * 256 >= slice.row_count) break;
    uint32_t row[BATCH], r[BATCH], valid = 0;
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      r[i] = (block * BATCH + i) * 256 + threadIdx.x;
      if (r[i] < slice.row_count) valid |= 1u << i;
      row[i] = slice.row_begin + (r[i] < slice.row_count ? r[i] : 0);
    }
    uint64_t bits[BATCH];
    uint32_t nulls = 0;
    decode_rows<BATCH>(s, a.segments, slice.chunk, row, valid, bits, &nulls);
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      if (!((valid >> i) & 1)) continue;
      const uint64_t at = base + r[i];
      const bool is_null = (nulls >> i) & 1;
      if (a.width == 4) {
        uint32_t word = static_cast<uint32_t>(bits[i]);
        if (a.is_float) word = __float_as_uint(static_cast<float>(__longlong_as_double(static_cast<long long>(bits[i]))));
        static_cast<uint32_t*>(a.values)[at] = is_null ? 0u : word;
      } else {
        static_cast<uint64_t*>(a.values)[at] = is_null ? 0ull : bits[i];
      }
      if (a.nulls) a.nulls[at] = is_null ? 1 : 0;
    }
  }
}

// The same for a REFERENCE column (an operator result read through its PosLists: the foreign key of a join result, the columns an
// aggregate reads behind three joins).  decode_rows takes such rows one at a time -- RowID, then the referenced segment's descriptor, then
// its NULL word, its vector, its dictionary or block minimum: five dependent loads a row, nothing in flight beside them (36 M foreign keys
// of SSB Q4.1: 432 us).  Here eight rows per lane go through every level together: the RowIDs, then the descriptors' words, then the
// first-level elements, then the second-level ones.
__global__ __launch_bounds__(256) void export_reference_rows(ExportArgs a) {
  const Slice slice = a.slices[blockIdx.x];
  if (slice.row_count == 0) return;
  const DevSegment s = a.segments[slice.chunk];
  const hy_row_id* pos = static_cast<const hy_row_id*>(s.data);
  const uint64_t base = a.row_base[slice.chunk] + slice.row_begin;
  constexpr int BATCH = 8;
#pragma unroll 1
  for (uint32_t block = 0; block * BATCH * 256 < slice.row_count; ++block) {
    uint32_t r[BATCH], chunk[BATCH], offset[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      r[i] = (block * BATCH + i) * 256 + threadIdx.x;
      chunk[i] = s.ref_chunk_id;
      offset[i] = 0xFFFFFFFFu;
      if (r[i] < slice.row_count) {
        if (pos) { const hy_row_id rid = pos[slice.row_begin + r[i]]; chunk[i] = rid.chunk_id; offset[i] = rid.chunk_offset; }
        else offset[i] = slice.row_begin + r[i];   // (EntireChunkPosList)
      }
    }
    const void* data[BATCH];
    const void* aux[BATCH];
    const uint64_t* null_words[BATCH];
    uint32_t aux_size[BATCH], kind[BATCH];   // kind: encoding | data_type << 8 | width << 16
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      data[i] = aux[i] = nullptr;
      null_words[i] = nullptr;
      aux_size[i] = kind[i] = 0;
      if (offset[i] == 0xFFFFFFFFu) continue;   // NULL_ROW_ID (or past the slice)
      const DevSegment* b = s.ref + chunk[i];
      data[i] = b->data;
      aux[i] = b->aux;
      null_words[i] = b->nulls;
      aux_size[i] = b->aux_size;
      kind[i] = static_cast<uint32_t>(b->encoding) | static_cast<uint32_t>(b->data_type) << 8 | static_cast<uint32_t>(b->width) << 16;
    }
    uint64_t first[BATCH], null_word[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      first[i] = null_word[i] = 0;
      if (offset[i] == 0xFFFFFFFFu) continue;
      const uint32_t encoding = kind[i] & 0xFF, type = (kind[i] >> 8) & 0xFF, width = kind[i] >> 16;
      if (encoding == HY_ENC_DICTIONARY || encoding == HY_ENC_FRAME_OF_REFERENCE) first[i] = aload_compressed(data[i], width, offset[i]);
      else if (type == HY_TYPE_INT || type == HY_TYPE_FLOAT) first[i] = static_cast<const uint32_t*>(data[i])[offset[i]];
      else first[i] = static_cast<const uint64_t*>(data[i])[offset[i]];
      if (encoding != HY_ENC_DICTIONARY && null_words[i]) null_word[i] = null_words[i][offset[i] >> 6];
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      if (r[i] >= slice.row_count) continue;
      const uint32_t encoding = kind[i] & 0xFF, type = (kind[i] >> 8) & 0xFF;
      bool is_null = offset[i] == 0xFFFFFFFFu || ((null_word[i] >> (offset[i] & 63)) & 1);
      uint64_t word = first[i];   // the value's own bits: 4 bytes of an int32 / float, 8 of an int64 / double
      if (!is_null && encoding == HY_ENC_DICTIONARY) {
        if (first[i] >= aux_size[i]) is_null = true;
        else if (type == HY_TYPE_INT || type == HY_TYPE_FLOAT) word = static_cast<const uint32_t*>(aux[i])[first[i]];
        else word = static_cast<const uint64_t*>(aux[i])[first[i]];
      } else if (!is_null && encoding == HY_ENC_FRAME_OF_REFERENCE) {
        word = static_cast<uint32_t>(first[i]) + static_cast<uint32_t>(static_cast<const int32_t*>(aux[i])[offset[i] / HY_FOR_BLOCK_SIZE]);
      }
      const uint64_t at = base + r[i];
      if (a.width == 4) static_cast<uint32_t*>(a.values)[at] = is_null ? 0u : static_cast<uint32_t>(word);
      else static_cast<uint64_t*>(a.values)[at] = is_null ? 0ull : word;
      if (a.nulls) a.nulls[at] = is_null ? 1 : 0;
    }
  }
}

// ---- hash repartition ------------------------------------------------------------------------------------------------------
constexpr uint32_t MAX_PARTS = 16;
struct RepartitionArgs {
  const DevSegment* segments;
  const Slice* slices;
  uint32_t n_slices;
  uint32_t parts;
  uint32_t chunk_id_offset;       // added to the chunk ids of the RowIDs that travel (the shard's first chunk in the whole table)
  uint32_t key_width;             // 4: int32 keys travel as int32, 8: int64
  uint32_t* slice_counts;         // [n_slices][parts]
  const uint64_t* slice_offsets;  // pack: [parts][n_slices] flattened destination-major exclusive scan (+ total)
  void* keys_out;
  hy_row_id* rows_out;
};

__device__ __forceinline__ bool repartition_key(const RepartitionArgs& a, const DevSegment& s, uint32_t chunk, uint32_t row, int64_t* key) {
  const Value v = column_value(a.segments, chunk, row);
  *key = v.i;
  return !v.is_null;   // NULL keys never find a partner: they stay home (Inner / Semi joins)
}

// One workgroup per slice, rows visited as row = k * 256 + tid so that compaction order is row order.  MODE 0 counts the
// slice's rows per destination, MODE 1 writes them behind the scanned offsets.
template <int MODE>
__global__ __launch_bounds__(256) void repartition_rows(RepartitionArgs a) {
  __shared__ uint32_t s_count[32][4][MAX_PARTS];   // [k][wave][destination]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Slice slice = a.slices[blockIdx.x];
  const DevSegment s = a.segments[slice.chunk];
#pragma unroll 1
  for (uint32_t k = 0; k < 32; ++k) {
    const uint32_t r = k * 256 + tid;
    uint32_t dest = 0;   // + 1
    if (r < slice.row_count) {
      int64_t key;
      if (repartition_key(a, s, slice.chunk, slice.row_begin + r, &key)) dest = static_cast<uint32_t>(static_cast<uint64_t>(key) % a.parts) + 1;
    }
    for (uint32_t d = 0; d < a.parts; ++d) {
      const uint64_t mask = __ballot(dest == d + 1);
      if (lane == 0) s_count[k][wave][d] = __popcll(mask);
    }
  }
  __syncthreads();
  if (MODE == 0) {
    if (tid < a.parts) {
      uint32_t sum = 0;
      for (uint32_t k = 0; k < 32; ++k)
        for (uint32_t w = 0; w < 4; ++w) sum += s_count[k][w][tid];
      a.slice_counts[static_cast<size_t>(blockIdx.x) * a.parts + tid] = sum;
    }
    return;
  }
  if (tid < a.parts) {   // exclusive prefix over (k, wave) for this destination
    uint32_t run = 0;
    for (uint32_t k = 0; k < 32; ++k)
      for (uint32_t w = 0; w < 4; ++w) { const uint32_t c = s_count[k][w][tid]; s_count[k][w][tid] = run; run += c; }
  }
  __syncthreads();
#pragma unroll 1
  for (uint32_t k = 0; k < 32; ++k) {
    const uint32_t r = k * 256 + tid;
    uint32_t dest = 0;   // + 1 (the key is read again: cheaper than keeping 32 destinations per thread)
    int64_t key = 0;
    if (r < slice.row_count && repartition_key(a, s, slice.chunk, slice.row_begin + r, &key)) dest = static_cast<uint32_t>(static_cast<uint64_t>(key) % a.parts) + 1;
    for (uint32_t d = 0; d < a.parts; ++d) {
      const uint64_t mask = __ballot(dest == d + 1);
      if (dest != d + 1) continue;
      const uint64_t pos = a.slice_offsets[static_cast<size_t>(d) * a.n_slices + blockIdx.x] + s_count[k][wave][d] + __popcll(mask & ((1ull << lane) - 1));
      if (a.key_width == 4) static_cast<int32_t*>(a.keys_out)[pos] = static_cast<int32_t>(key);
      else static_cast<int64_t*>(a.keys_out)[pos] = key;
      a.rows_out[pos] = hy_row_id{slice.chunk + a.chunk_id_offset, slice.row_begin + r};
    }
  }
}

// slice_counts [n_slices][parts] -> offsets [parts][n_slices] (destination-major exclusive scan), totals[parts]; one workgroup
// (shards have a few thousand slices).
__global__ __launch_bounds__(1024) void repartition_scan(const uint32_t* slice_counts, uint32_t n_slices, uint32_t parts, uint64_t* offsets, uint64_t* totals) {
  __shared__ uint64_t s_partial[1024];
  __shared__ uint64_t s_base;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const uint32_t per = (n_slices + 1023) / 1024;
  for (uint32_t d = 0; d < parts; ++d) {
    const uint32_t begin = tid * per < n_slices ? tid * per : n_slices, end = begin + per < n_slices ? begin + per : n_slices;
    uint64_t sum = 0;
    for (uint32_t i = begin; i < end; ++i) sum += slice_counts[static_cast<size_t>(i) * parts + d];
    s_partial[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint64_t run = s_base;
      for (uint32_t i = 0; i < 1024; ++i) { const uint64_t v = s_partial[i]; s_partial[i] = run; run += v; }
      totals[d] = run - s_base;
      s_base = run;
    }
    __syncthreads();
    uint64_t run = s_partial[tid];
    for (uint32_t i = begin; i < end; ++i) { offsets[static_cast<size_t>(d) * n_slices + i] = run; run += slice_counts[static_cast<size_t>(i) * parts + d]; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gather_row_ids(const hy_row_id* table, uint64_t table_rows, uint32_t chunk_rows, const hy_row_id* positions, uint64_t n, hy_row_id* out) {
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const hy_row_id p = positions[i];
    hy_row_id r{0xFFFFFFFFu, 0xFFFFFFFFu};
    if (p.chunk_offset != 0xFFFFFFFFu) {
      const uint64_t at = static_cast<uint64_t>(p.chunk_id) * chunk_rows + p.chunk_offset;
      if (at < table_rows) r = table[at];
    }
    out[i] = r;
  }
}

// The same, four positions per thread and step as two 16-byte loads and stores (both arrays on 16-byte boundaries): 36 M positions of an
// SSB join result took 275 us one RowID at a time -- 2.1 TB/s for an access pattern that is two streams and a small table.
typedef uint32_t gather_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gather_row_ids_wide(const hy_row_id* table, uint64_t table_rows, uint32_t chunk_rows, const hy_row_id* positions, uint64_t n, hy_row_id* out) {
  const gather_u32x4* in = reinterpret_cast<const gather_u32x4*>(positions);
  gather_u32x4* wide_out = reinterpret_cast<gather_u32x4*>(out);
  const uint64_t n_vectors = n / 2;
  auto lookup = [&](uint32_t chunk_id, uint32_t chunk_offset) -> hy_row_id {
    hy_row_id r{0xFFFFFFFFu, 0xFFFFFFFFu};
    if (chunk_offset != 0xFFFFFFFFu) {
      const uint64_t at = static_cast<uint64_t>(chunk_id) * chunk_rows + chunk_offset;
      if (at < table_rows) r = table[at];
    }
    return r;
  };
  for (uint64_t v = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) * 2; v < n_vectors; v += static_cast<uint64_t>(gridDim.x) * 512) {
    const bool second = v + 1 < n_vectors;
    const gather_u32x4 p0 = __builtin_nontemporal_load(in + v);
    const gather_u32x4 p1 = second ? __builtin_nontemporal_load(in + v + 1) : gather_u32x4{0, 0xFFFFFFFFu, 0, 0xFFFFFFFFu};
    const hy_row_id a0 = lookup(p0.x, p0.y), a1 = lookup(p0.z, p0.w), b0 = lookup(p1.x, p1.y), b1 = lookup(p1.z, p1.w);
    __builtin_nontemporal_store(gather_u32x4{a0.chunk_id, a0.chunk_offset, a1.chunk_id, a1.chunk_offset}, wide_out + v);
    if (second) __builtin_nontemporal_store(gather_u32x4{b0.chunk_id, b0.chunk_offset, b1.chunk_id, b1.chunk_offset}, wide_out + v + 1);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
    const hy_row_id p = positions[n - 1];
    out[n - 1] = lookup(p.chunk_id, p.chunk_offset);
  }
}

// positions (c, o) of a reference table -> the RowIDs its pos lists hold there: write_output_segments' dereferencing of a reference input
// (join_output_writing.cpp:95-200: `(*input_pos_list)[row.chunk_offset]` per pair, NULL RowIDs stay NULL), two positions per thread and step.
__global__ __launch_bounds__(256) void gather_through_pos_lists(const DevSegment* segments, uint32_t n_chunks, const hy_row_id* positions, uint64_t n, hy_row_id* out) {
  auto lookup = [&](uint32_t chunk_id, uint32_t chunk_offset) -> hy_row_id {
    hy_row_id r{0xFFFFFFFFu, 0xFFFFFFFFu};
    if (chunk_offset == 0xFFFFFFFFu || chunk_id >= n_chunks) return r;
    const DevSegment& s = segments[chunk_id];
    if (chunk_offset >= s.size) return r;
    if (!s.data) return hy_row_id{s.ref_chunk_id, chunk_offset};   // EntireChunkPosList
    return static_cast<const hy_row_id*>(s.data)[chunk_offset];
  };
  const bool wide = reinterpret_cast<uintptr_t>(positions) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (wide) {
    const gather_u32x4* in = reinterpret_cast<const gather_u32x4*>(positions);
    gather_u32x4* wide_out = reinterpret_cast<gather_u32x4*>(out);
    for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; v < n / 2; v += static_cast<uint64_t>(gridDim.x) * 256) {
      const gather_u32x4 p = __builtin_nontemporal_load(in + v);
      const hy_row_id a = lookup(p.x, p.y), b = lookup(p.z, p.w);
      __builtin_nontemporal_store(gather_u32x4{a.chunk_id, a.chunk_offset, b.chunk_id, b.chunk_offset}, wide_out + v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) out[n - 1] = lookup(positions[n - 1].chunk_id, positions[n - 1].chunk_offset);
  } else {
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) out[i] = lookup(positions[i].chunk_id, positions[i].chunk_offset);
  }
}

}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_column_export(const hy_column* column, void* values, uint8_t* nulls) { return hy::export_column_at(column, values, nulls, nullptr); }

}  // extern "C"

namespace hy {
// hy_column_export with the chunks' first rows given (device array [n_chunks], in elements): row r of chunk c goes to values[row_base[c] + r]
// -- join.hip materialises reference inputs chunk by chunk on 16-byte boundaries.  nullptr: the column's own (back to back).
hy_status export_column_at(const hy_column* column, void* values, uint8_t* nulls, const uint64_t* d_row_base) {
  if (!column || !values) return fail(HY_ERR_INVALID, "hy_column_export: null argument");
  HY_TRY(on_this_device(column, "hy_column_export"));
  if (column->is_mvcc) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
  if (column->data_type < HY_TYPE_INT || column->data_type > HY_TYPE_DOUBLE) return fail(HY_ERR_UNSUPPORTED, "hy_column_export: numeric columns only");
  if (column->has_dictionary_without_values) return fail(HY_ERR_UNSUPPORTED, "hy_column_export: the dictionary values are not on the device");
  if (!column->n_slices || !column->rows) return HY_OK;
  HY_TRY(plain_column(column, &column));
  hipStream_t stream = current_stream();
  ExportArgs a{};
  a.segments = column->d_segments;
  a.slices = column->d_slices;
  a.row_base = d_row_base ? d_row_base : column->d_row_base;
  a.values = values;
  a.nulls = nulls;
  a.width = (column->data_type == HY_TYPE_INT || column->data_type == HY_TYPE_FLOAT) ? 4 : 8;
  a.is_float = (column->data_type == HY_TYPE_FLOAT || column->data_type == HY_TYPE_DOUBLE) ? 1 : 0;
  if (column->is_reference) hipLaunchKernelGGL(export_reference_rows, dim3(column->n_slices), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(export_rows, dim3(column->n_slices), dim3(256), 0, stream, a);
  HY_HIP(hipGetLastError());
  return HY_OK;
}
}  // namespace hy

extern "C" {

static hy_status repartition_check(const hy_column* column, uint32_t parts) {
  if (!column) return fail(HY_ERR_INVALID, "hy_repartition: null column");
  HY_TRY(on_this_device(column, "hy_repartition"));
  if (parts == 0 || parts > MAX_PARTS) return fail(HY_ERR_INVALID, "hy_repartition: 1..%u destinations", MAX_PARTS);
  if (column->is_mvcc) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
  if (column->data_type != HY_TYPE_INT && column->data_type != HY_TYPE_LONG) return fail(HY_ERR_UNSUPPORTED, "hy_repartition: integer join keys (std::hash of a float is not its value)");
  if (column->has_dictionary_without_values) return fail(HY_ERR_UNSUPPORTED, "hy_repartition: the dictionary values are not on the device");
  return HY_OK;
}

hy_status hy_repartition_count(const hy_column* column, uint32_t parts, uint64_t* counts) {
  HY_TRY(repartition_check(column, parts));
  HY_TRY(plain_column(column, &column));
  if (!counts) return fail(HY_ERR_INVALID, "hy_repartition_count: counts missing");
  for (uint32_t d = 0; d < parts; ++d) counts[d] = 0;
  if (!column->n_slices || !column->rows) return HY_OK;
  hipStream_t stream = current_stream();
  DeviceBuffer slice_counts, offsets, totals;
  HY_TRY(slice_counts.alloc(4 * size_t{column->n_slices} * parts));
  HY_TRY(offsets.alloc(8 * size_t{column->n_slices} * parts));
  HY_TRY(totals.alloc(8 * size_t{MAX_PARTS}));
  RepartitionArgs a{};
  a.segments = column->d_segments;
  a.slices = column->d_slices;
  a.n_slices = column->n_slices;
  a.parts = parts;
  a.slice_counts = slice_counts.as<uint32_t>();
  hipLaunchKernelGGL(repartition_rows<0>, dim3(column->n_slices), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(repartition_scan, dim3(1), dim3(1024), 0, stream, slice_counts.as<uint32_t>(), column->n_slices, parts, offsets.as<uint64_t>(), totals.as<uint64_t>());
  HY_HIP(hipMemcpyAsync(counts, totals.ptr, 8 * size_t{parts}, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  return HY_OK;
}

hy_status hy_repartition_pack(const hy_column* column, uint32_t parts, uint32_t chunk_id_offset, void* keys_out, hy_row_id* row_ids_out, uint64_t capacity,
                              uint64_t* counts) {
  HY_TRY(repartition_check(column, parts));
  HY_TRY(plain_column(column, &column));
  if (!counts || (capacity && (!keys_out || !row_ids_out))) return fail(HY_ERR_INVALID, "hy_repartition_pack: output buffer missing");
  for (uint32_t d = 0; d < parts; ++d) counts[d] = 0;
  if (!column->n_slices || !column->rows) return HY_OK;
  hipStream_t stream = current_stream();
  DeviceBuffer slice_counts, offsets, totals;
  HY_TRY(slice_counts.alloc(4 * size_t{column->n_slices} * parts));
  HY_TRY(offsets.alloc(8 * size_t{column->n_slices} * parts));
  HY_TRY(totals.alloc(8 * size_t{MAX_PARTS}));
  RepartitionArgs a{};
  a.segments = column->d_segments;
  a.slices = column->d_slices;
  a.n_slices = column->n_slices;
  a.parts = parts;
  a.chunk_id_offset = chunk_id_offset;
  a.key_width = column->data_type == HY_TYPE_INT ? 4 : 8;
  a.slice_counts = slice_counts.as<uint32_t>();
  a.slice_offsets = offsets.as<uint64_t>();
  a.keys_out = keys_out;
  a.rows_out = row_ids_out;
  hipLaunchKernelGGL(repartition_rows<0>, dim3(column->n_slices), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(repartition_scan, dim3(1), dim3(1024), 0, stream, slice_counts.as<uint32_t>(), column->n_slices, parts, offsets.as<uint64_t>(), totals.as<uint64_t>());
  HY_HIP(hipMemcpyAsync(counts, totals.ptr, 8 * size_t{parts}, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  uint64_t total = 0;
  for (uint32_t d = 0; d < parts; ++d) total += counts[d];
  if (total > capacity) return fail(HY_ERR_CAPACITY, "hy_repartition_pack: %llu tuples, capacity %llu", static_cast<unsigned long long>(total), static_cast<unsigned long long>(capacity));
  hipLaunchKernelGGL(repartition_rows<1>, dim3(column->n_slices), dim3(256), 0, stream, a);
  HY_HIP(hipGetLastError());
  return HY_OK;
}

hy_status hy_gather_row_ids(const hy_row_id* table, uint64_t table_rows, uint32_t chunk_rows, const hy_row_id* positions, uint64_t n, hy_row_id* out) {
  // (an empty table -- a rank that received no tuples of the other side -- has no buffer: every position then is out of range and yields the NULL RowID)
  if (n && ((!table && table_rows) || !positions || !out || !chunk_rows)) return fail(HY_ERR_INVALID, "hy_gather_row_ids: null argument");
  if (!n) return HY_OK;
  const bool wide = n >= 4096 && reinterpret_cast<uintptr_t>(positions) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n / (wide ? 4 : 1) + 255) / 256, 16384));
  if (wide) hipLaunchKernelGGL(gather_row_ids_wide, dim3(grid), dim3(256), 0, current_stream(), table, table_rows, chunk_rows, positions, n, out);
  else hipLaunchKernelGGL(gather_row_ids, dim3(grid), dim3(256), 0, current_stream(), table, table_rows, chunk_rows, positions, n, out);
  HY_HIP(hipGetLastError());
  return HY_OK;
}

hy_status hy_poslist_gather(const hy_column* reference, const hy_row_id* positions, uint64_t n, hy_row_id* out) {
  if (!reference || (n && (!positions || !out))) return fail(HY_ERR_INVALID, "hy_poslist_gather: null argument");
  HY_TRY(on_this_device(reference, "hy_poslist_gather"));
  if (!reference->is_reference) return fail(HY_ERR_INVALID, "hy_poslist_gather: a column of reference segments is needed (a data table's positions ARE its RowIDs)");
  if (!n) return HY_OK;
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n / 2 + 255) / 256 + 1, 16384));
  hipLaunchKernelGGL(gather_through_pos_lists, dim3(grid), dim3(256), 0, current_stream(), reference->d_segments, reference->n_chunks, positions, n, out);
  HY_HIP(hipGetLastError());
  return HY_OK;
}

}  // extern "C"
