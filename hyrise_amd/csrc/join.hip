// join.hip -- JoinHash on MI355X: equi-joins of int32/int64 columns, all JoinHash modes except secondary predicates.
//
// What it replaces (reference, CPU):
//   JoinHash::_on_execute / JoinHashImpl::_on_execute                 operators/join_hash.cpp:116-225, 270-572
//   materialize_input / partition_by_radix / build / probe(_semi_anti) operators/join_hash/join_hash_steps.hpp:274-922
//
// The reference's output order is fully determined by its algorithm: pairs come radix partition by radix partition
// (low `radix_bits` bits of the key; `radix_bits` == 0: probe chunk by probe chunk), inside a partition by probe row,
// inside a probe row by build row (hash-table insertion order == (chunk, row) order), and every 131 070 materialised
// probe elements of a partition start a new output PosList.  The device code produces exactly that order without
// materialising the partitions themselves:
//   build side   materialise (key, RowID) in row order (count / write passes with ballot compaction), radix-sort it
//                by key only if it is not already sorted (a primary-key column usually is), and lay an
//                order-preserving bucket directory over the sorted keys: bucket = (key - min) >> shift, open-ended
//                runs inside a bucket are resolved by a short binary search.  Equal keys are adjacent in build-row
//                order, so a probe hit is just (start, count) into the sorted RowID array.
//   probe side   two passes over the (still encoded) probe column, 2048-row tiles:
//                1. histogram: per tile and partition, the number of materialised probe elements and of output pairs
//                2. scatter: exclusive prefix sums of those histograms give every (partition, tile) its output range;
//                   inside a tile a wave-level match-any ranking (ballots over the radix bits) keeps the order stable,
//                   and every lane writes its pairs straight to their final positions.
// HBM traffic: build keys + 2 x probe keys + 16 B per pair (+ the small histograms); no 12-byte PartitionedElement
// arrays are ever written.  The Bloom filters of the reference are reproduced only where they are observable: the
// build side's filter decides which probe elements count as "materialised" (it shifts the 131 070-element cuts).
#include "hy_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace hy {

constexpr uint32_t JOIN_TILE = 2048;                 // rows per workgroup tile (8 per lane)
constexpr uint32_t PROBE_SIZE_PER_CHUNK = 65535u * 2u;  // join_hash_steps.hpp:47
constexpr uint32_t BLOOM_BITS = 1u << 20;            // join_hash_steps.hpp:252
// The device keeps the filter as one BYTE per bit: setting a bit is a plain store (no atomics, races are benign), and
// the 1 MiB array stays L2-resident for the probe.
constexpr uint32_t INVALID_PARTITION = 0x1FF;

// ---- decoding ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t jload_compressed(const void* data, uint32_t width, uint32_t i) {
  if (width == 1) return static_cast<const uint8_t*>(data)[i];
  if (width == 2) return static_cast<const uint16_t*>(data)[i];
  return static_cast<const uint32_t*>(data)[i];
}

// int32/int64 value of row `row` of a DATA segment; returns true if NULL.
__device__ __forceinline__ bool data_key(const DevSegment& s, uint32_t row, int64_t* key) {
  *key = 0;
  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = jload_compressed(s.data, s.width, row);
    if (vid >= s.aux_size) return true;
    *key = s.data_type == HY_TYPE_INT ? static_cast<int64_t>(static_cast<const int32_t*>(s.aux)[vid]) : static_cast<const int64_t*>(s.aux)[vid];
    return false;
  }
  if (s.nulls && ((s.nulls[row >> 6] >> (row & 63)) & 1)) return true;
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    *key = static_cast<int32_t>(jload_compressed(s.data, s.width, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]));
    return false;
  }
  *key = s.data_type == HY_TYPE_INT ? static_cast<int64_t>(static_cast<const int32_t*>(s.data)[row]) : static_cast<const int64_t*>(s.data)[row];
  return false;
}

__device__ __forceinline__ bool column_key(const DevSegment* segments, uint32_t chunk, uint32_t row, int64_t* key) {
  const DevSegment& s = segments[chunk];
  if (s.encoding != HY_ENC_REFERENCE) return data_key(s, row, key);
  hy_row_id r;
  if (s.data) r = static_cast<const hy_row_id*>(s.data)[row];
  else { r.chunk_id = s.ref_chunk_id; r.chunk_offset = row; }
  *key = 0;
  if (r.chunk_offset == 0xFFFFFFFFu) return true;
  return data_key(s.ref[r.chunk_id], r.chunk_offset, key);
}

__device__ __forceinline__ bool bloom_test(const uint8_t* bloom, uint64_t hash) {
  return bloom[static_cast<uint32_t>(hash) & (BLOOM_BITS - 1)] != 0;
}

// ---- build side: materialise -------------------------------------------------------------------------------------------
// One workgroup per 8192-row slice; rows are visited as row = k * 256 + tid (k = 0..31) so that compaction order is
// row order.  MODE 0 counts, MODE 1 writes.
struct MaterializeArgs {
  const DevSegment* segments;
  const Slice* slices;
  uint32_t n_slices;
  uint32_t keep_nulls;
  const uint8_t* bloom_in;        // nullptr = every bit set
  uint8_t* bloom_out;             // may be nullptr
  uint32_t* slice_counts;         // [n_slices]
  const uint64_t* slice_offsets;  // MODE 1
  uint64_t* keys;                 // MODE 1: sign-extended key bits
  hy_row_id* row_ids;             // MODE 1
  uint32_t* any_null;             // set to 1 if a NULL was materialised (AntiNullAsTrue early-out)
};

template <int MODE>
__global__ __launch_bounds__(256) void join_materialize(MaterializeArgs a) {
  __shared__ uint32_t s_count[32][4];
  __shared__ uint32_t s_offset[32][4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Slice slice = a.slices[blockIdx.x];
  uint32_t keep_bits = 0;
  for (uint32_t k = 0; k < 32; ++k) {
    const uint32_t r = k * 256 + tid;
    bool keep = false;
    if (r < slice.row_count) {
      int64_t key;
      const bool is_null = column_key(a.segments, slice.chunk, slice.row_begin + r, &key);
      keep = !is_null || a.keep_nulls;
      if (keep && !is_null && a.bloom_in && !a.keep_nulls) keep = bloom_test(a.bloom_in, static_cast<uint64_t>(key));
    }
    const uint64_t ballot = __ballot(keep);
    if (lane == 0) s_count[k][wave] = __popcll(ballot);
    keep_bits |= (keep ? 1u : 0u) << k;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t sum = 0;
    for (uint32_t k = 0; k < 32; ++k)
      for (uint32_t w = 0; w < 4; ++w) { s_offset[k][w] = sum; sum += s_count[k][w]; }
    if (MODE == 0) a.slice_counts[blockIdx.x] = sum;
  }
  if (MODE == 0) return;
  __syncthreads();
  const uint64_t base = a.slice_offsets[blockIdx.x];
  for (uint32_t k = 0; k < 32; ++k) {
    const bool keep = (keep_bits >> k) & 1;
    const uint64_t ballot = __ballot(keep);
    if (keep) {
      const uint32_t r = k * 256 + tid;
      int64_t key;
      const bool is_null = column_key(a.segments, slice.chunk, slice.row_begin + r, &key);
      const uint64_t pos = base + s_offset[k][wave] + __popcll(ballot & ((1ull << lane) - 1));
      a.keys[pos] = static_cast<uint64_t>(key);
      a.row_ids[pos] = hy_row_id{slice.chunk, slice.row_begin + r};
      if (is_null && a.any_null) *a.any_null = 1;
      if (a.bloom_out) a.bloom_out[static_cast<uint32_t>(key) & (BLOOM_BITS - 1)] = 1;
    }
  }
}

// Single-workgroup exclusive scan of u32 counts into u64 offsets (n is a few thousand); offsets[n] = total.
__global__ __launch_bounds__(1024) void scan_counts(const uint32_t* counts, uint64_t* offsets, uint32_t n) {
  __shared__ uint64_t s_partial[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t begin = tid * per, end = begin + per < n ? begin + per : n;
  uint64_t sum = 0;
  for (uint32_t i = begin; i < end; ++i) sum += counts[i];
  s_partial[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    uint64_t run = 0;
    for (uint32_t i = 0; i < 1024; ++i) { const uint64_t v = s_partial[i]; s_partial[i] = run; run += v; }
    offsets[n] = run;
  }
  __syncthreads();
  uint64_t run = s_partial[tid];
  for (uint32_t i = begin; i < end; ++i) { offsets[i] = run; run += counts[i]; }
}

// keys sorted ascending (unsigned bit order)?  Also OR-reduces all keys (significant bytes for the radix sort):
// one atomic per 1024-thread workgroup.
__global__ __launch_bounds__(1024) void check_sorted(const uint64_t* keys, uint64_t n, uint32_t* unsorted, unsigned long long* key_or) {
  __shared__ uint64_t s_bits[16];
  __shared__ uint32_t s_unsorted;
  if (threadIdx.x == 0) s_unsorted = 0;
  __syncthreads();
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  uint64_t bits = 0;
  if (i < n) {
    bits = keys[i];
    if (i + 1 < n && bits > keys[i + 1]) s_unsorted = 1;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bits |= __shfl_xor(bits, d, 64);
  if ((threadIdx.x & 63) == 0) s_bits[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t all = 0;
    for (uint32_t w = 0; w < 16; ++w) all |= s_bits[w];
    if (all) atomicOr(key_or, static_cast<unsigned long long>(all));
    if (s_unsorted) *unsorted = 1;
  }
}

// ---- stable LSD radix sort of (key, RowID) pairs, 8 bits per pass ---------------------------------------------------
// histogram: [256][tiles] (digit-major so that one exclusive scan over the flat array yields scatter bases)
constexpr uint32_t SORT_TILE = 2048;
__global__ __launch_bounds__(256) void sort_histogram(const uint64_t* keys, uint64_t n, uint32_t shift, uint32_t* hist, uint32_t n_tiles) {
  __shared__ uint32_t s_hist[256];
  const uint32_t tid = threadIdx.x;
  s_hist[tid] = 0;
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SORT_TILE;
  for (uint32_t k = 0; k < SORT_TILE / 256; ++k) {
    const uint64_t i = base + k * 256 + tid;
    if (i < n) atomicAdd(&s_hist[(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  hist[static_cast<size_t>(tid) * n_tiles + blockIdx.x] = s_hist[tid];
}

// Wave-level match-any on an 8-bit digit: mask of the lanes (among `valid`) holding the same digit.
__device__ __forceinline__ uint64_t match_any(uint32_t digit, bool valid, uint32_t bits) {
  uint64_t peers = __ballot(valid);
  for (uint32_t b = 0; b < bits; ++b) {
    const bool bit = (digit >> b) & 1;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return valid ? peers : 0;
}

__global__ __launch_bounds__(256) void sort_scatter(const uint64_t* keys_in, const hy_row_id* rows_in, uint64_t* keys_out, hy_row_id* rows_out,
                                                    uint64_t n, uint32_t shift, const uint64_t* bases, uint32_t n_tiles) {
  __shared__ uint32_t s_wave_hist[4][256];   // per-wave digit counts, then per-wave running positions (relative to the tile's base)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t w = 0; w < 4; ++w) s_wave_hist[w][tid] = 0;
  __syncthreads();
  const uint64_t tile_base = static_cast<uint64_t>(blockIdx.x) * SORT_TILE;
  // wave w owns elements [w*512, (w+1)*512) of the tile, 8 rounds of 64 consecutive elements
  uint32_t digits[8];
  for (uint32_t round = 0; round < 8; ++round) {
    const uint64_t i = tile_base + wave * 512 + round * 64 + lane;
    const bool valid = i < n;
    const uint32_t digit = valid ? static_cast<uint32_t>((keys_in[i] >> shift) & 0xFF) : 0;
    digits[round] = digit;
    const uint64_t peers = match_any(digit, valid, 8);
    if (valid && (peers >> lane) >> 1 == 0) atomicAdd(&s_wave_hist[wave][digit], static_cast<uint32_t>(__popcll(peers)));  // highest peer adds
  }
  __syncthreads();
  {  // thread = digit: exclusive prefix over the 4 waves
    uint32_t run = 0;
    for (uint32_t w = 0; w < 4; ++w) { const uint32_t v = s_wave_hist[w][tid]; s_wave_hist[w][tid] = run; run += v; }
  }
  __syncthreads();
  for (uint32_t round = 0; round < 8; ++round) {
    const uint64_t i = tile_base + wave * 512 + round * 64 + lane;
    const bool valid = i < n;
    const uint32_t digit = digits[round];
    const uint64_t peers = match_any(digit, valid, 8);
    if (valid) {
      const uint32_t rank = __popcll(peers & ((1ull << lane) - 1));
      const uint32_t local = s_wave_hist[wave][digit] + rank;
      const uint64_t pos = bases[static_cast<size_t>(digit) * n_tiles + blockIdx.x] + local;
      keys_out[pos] = keys_in[i];
      rows_out[pos] = rows_in[i];
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers >> lane) >> 1 == 0) s_wave_hist[wave][digit] += static_cast<uint32_t>(__popcll(peers));
    __builtin_amdgcn_wave_barrier();
  }
}

// Exclusive scan of a long u32 array into u64: per-block sums (4096 elements per workgroup), a single-workgroup scan
// of the block sums, then per-block scans with the block's offset.  out[n] = total.
constexpr uint32_t SCAN_BLOCK = 4096;
__global__ __launch_bounds__(256) void scan_block_sums(const uint32_t* in, uint64_t n, uint64_t* block_sums) {
  __shared__ uint64_t s_wave[4];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SCAN_BLOCK;
  uint64_t sum = 0;
  for (uint32_t k = 0; k < SCAN_BLOCK / 256; ++k) {
    const uint64_t i = base + k * 256 + threadIdx.x;
    if (i < n) sum += in[i];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

__global__ __launch_bounds__(1024) void scan_block_offsets(uint64_t* block_sums, uint32_t n_blocks, uint64_t* total_out) {
  __shared__ uint64_t s_partial[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n_blocks + 1023) / 1024;
  const uint32_t begin = tid * per < n_blocks ? tid * per : n_blocks, end = begin + per < n_blocks ? begin + per : n_blocks;
  uint64_t sum = 0;
  for (uint32_t i = begin; i < end; ++i) sum += block_sums[i];
  s_partial[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    uint64_t run = 0;
    for (uint32_t i = 0; i < 1024; ++i) { const uint64_t v = s_partial[i]; s_partial[i] = run; run += v; }
    *total_out = run;
  }
  __syncthreads();
  uint64_t run = s_partial[tid];
  for (uint32_t i = begin; i < end; ++i) { const uint64_t v = block_sums[i]; block_sums[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void scan_blocks(const uint32_t* in, uint64_t n, const uint64_t* block_offsets, uint64_t* out) {
  __shared__ uint64_t s_wave[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SCAN_BLOCK + static_cast<uint64_t>(tid) * (SCAN_BLOCK / 256);
  uint32_t values[SCAN_BLOCK / 256];
  uint64_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_BLOCK / 256; ++k) {
    values[k] = base + k < n ? in[base + k] : 0;
    sum += values[k];
  }
  uint64_t inclusive = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(inclusive, d, 64);
    if (lane >= static_cast<uint32_t>(d)) inclusive += t;
  }
  if (lane == 63) s_wave[wave] = inclusive;
  __syncthreads();
  uint64_t run = block_offsets[blockIdx.x] + inclusive - sum;
  for (uint32_t w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
  for (uint32_t k = 0; k < SCAN_BLOCK / 256; ++k) {
    if (base + k < n) out[base + k] = run;
    run += values[k];
  }
}

// out[i] = src[index[i]]
__global__ void gather_u64(const uint64_t* src, const uint64_t* index, uint64_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[index[i]];
}

// ---- bucket directory over the sorted build keys ------------------------------------------------------------------------
struct Directory {
  const uint64_t* keys;      // sorted (unsigned order of the sign-extended bits)
  const hy_row_id* row_ids;  // same order
  const uint32_t* dir;       // [n_buckets + 1] first position of every bucket
  uint64_t n;
  uint64_t key_min;
  uint64_t key_max;
  uint32_t shift;
  uint32_t n_buckets;
};

__global__ void directory_fill(const uint64_t* keys, uint64_t n, uint64_t key_min, uint32_t shift, uint32_t n_buckets, uint32_t* dir) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t bucket = (keys[i] - key_min) >> shift;
  const int64_t previous = i == 0 ? -1 : static_cast<int64_t>((keys[i - 1] - key_min) >> shift);
  for (int64_t b = previous + 1; b <= static_cast<int64_t>(bucket); ++b) dir[b] = static_cast<uint32_t>(i);
  if (i + 1 == n) {
    for (uint64_t b = bucket + 1; b <= n_buckets; ++b) dir[b] = static_cast<uint32_t>(n);
  }
}

// (start, count) of `key` in the sorted build keys.
__device__ __forceinline__ void directory_lookup(const Directory& d, uint64_t key, uint32_t* start, uint32_t* count) {
  *start = 0;
  *count = 0;
  if (d.n == 0 || key < d.key_min || key > d.key_max) return;
  const uint64_t bucket = (key - d.key_min) >> d.shift;
  uint32_t lo = d.dir[bucket], hi = d.dir[bucket + 1];
  const uint32_t bucket_end = hi;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (d.keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  if (lo == bucket_end || d.keys[lo] != key) return;
  *start = lo;
  uint32_t end = lo + 1;
  if (end < bucket_end && d.keys[end] == key) {   // duplicates: upper bound inside the bucket
    uint32_t l = end, h = bucket_end;
    while (l < h) {
      const uint32_t mid = l + (h - l) / 2;
      if (d.keys[mid] <= key) l = mid + 1; else h = mid;
    }
    end = l;
  }
  *count = end - lo;
}

// ---- probe side ------------------------------------------------------------------------------------------------------
struct ProbeArgs {
  const DevSegment* segments;     // probe column
  const Slice* slices;            // 8192-row slices; a tile is a quarter of a slice
  uint32_t n_tiles;
  uint32_t mode;                  // HY_JOIN_*
  uint32_t radix_bits;
  uint32_t keep_nulls;            // probe side keeps NULLs (Left/Right/Anti*)
  uint32_t build_rows_zero;       // build table has no rows (AntiNullAsTrue special case)
  const uint8_t* build_bloom;     // filter applied to the probe side, or nullptr
  Directory dir;
  // pass 1 out / pass 2 in
  uint32_t* hist_elements;        // [P][n_tiles]  (P = 1 << radix_bits, or 1)
  uint32_t* hist_pairs;           // [P][n_tiles]
  const uint64_t* base_elements;  // exclusive scans of the above (flat, partition-major)
  const uint64_t* base_pairs;
  const uint64_t* partition_element_origin;  // [P] or per chunk: scanned element count at the partition's start
  const uint32_t* partition_slice_base;      // [P] (radix) or [n_chunks] (radix_bits == 0): first slice index
  hy_row_id* build_out;           // may be nullptr (Semi/Anti)
  hy_row_id* probe_out;
  uint64_t* slice_offsets;
};

struct ProbeRow {
  uint32_t partition;   // INVALID_PARTITION if not materialised
  uint32_t emit;        // output pairs
  uint32_t start;       // first build position (emit real partners) -- unused when null_partner
  bool null_partner;    // emit NULL_ROW_ID as the build side
};

__device__ __forceinline__ ProbeRow probe_row(const ProbeArgs& a, uint32_t chunk, uint32_t row) {
  ProbeRow out{INVALID_PARTITION, 0, 0, false};
  int64_t key;
  const bool is_null = column_key(a.segments, chunk, row, &key);
  if (is_null && !a.keep_nulls) return out;
  const uint64_t hash = static_cast<uint64_t>(key);
  if (!is_null && !a.keep_nulls && a.build_bloom && !bloom_test(a.build_bloom, hash)) return out;   // join_hash_steps.hpp:354-358
  out.partition = a.radix_bits ? static_cast<uint32_t>(hash & ((1u << a.radix_bits) - 1)) : 0;
  uint32_t start = 0, count = 0;
  if (!is_null) directory_lookup(a.dir, hash, &start, &count);
  out.start = start;
  switch (a.mode) {
    case HY_JOIN_INNER: out.emit = count; break;
    case HY_JOIN_LEFT:
    case HY_JOIN_RIGHT:
      if (is_null || count == 0) { out.emit = 1; out.null_partner = true; } else out.emit = count;
      break;
    case HY_JOIN_SEMI: out.emit = count > 0 ? 1 : 0; break;
    case HY_JOIN_ANTI_NULL_AS_FALSE: out.emit = (is_null || count == 0) ? 1 : 0; break;
    default:  // AntiNullAsTrue
      out.emit = is_null ? (a.build_rows_zero ? 1 : 0) : (count == 0 ? 1 : 0);
      break;
  }
  return out;
}

__device__ __forceinline__ void tile_rows(const ProbeArgs& a, uint32_t tile, uint32_t* chunk, uint32_t* row_begin, uint32_t* row_count) {
  const Slice slice = a.slices[tile >> 2];
  const uint32_t offset = (tile & 3) * JOIN_TILE;
  *chunk = slice.chunk;
  *row_begin = slice.row_begin + offset;
  *row_count = slice.row_count > offset ? (slice.row_count - offset < JOIN_TILE ? slice.row_count - offset : JOIN_TILE) : 0;
}

__global__ __launch_bounds__(256) void probe_histogram(ProbeArgs a) {
  __shared__ uint32_t s_elements[256];
  __shared__ uint32_t s_pairs[256];
  const uint32_t tid = threadIdx.x;
  const uint32_t partitions = 1u << a.radix_bits;
  s_elements[tid] = 0;
  s_pairs[tid] = 0;
  __syncthreads();
  uint32_t chunk, row_begin, row_count;
  tile_rows(a, blockIdx.x, &chunk, &row_begin, &row_count);
  for (uint32_t k = 0; k < JOIN_TILE / 256; ++k) {
    const uint32_t r = k * 256 + tid;
    if (r < row_count) {
      const ProbeRow p = probe_row(a, chunk, row_begin + r);
      if (p.partition != INVALID_PARTITION) {
        atomicAdd(&s_elements[p.partition], 1u);
        if (p.emit) atomicAdd(&s_pairs[p.partition], p.emit);
      }
    }
  }
  __syncthreads();
  if (tid < partitions) {
    a.hist_elements[static_cast<size_t>(tid) * a.n_tiles + blockIdx.x] = s_elements[tid];
    a.hist_pairs[static_cast<size_t>(tid) * a.n_tiles + blockIdx.x] = s_pairs[tid];
  }
}

__global__ __launch_bounds__(256) void probe_scatter(ProbeArgs a) {
  __shared__ uint16_t s_meta[JOIN_TILE];     // partition | null_partner << 15
  __shared__ uint32_t s_emit[JOIN_TILE];
  __shared__ uint32_t s_start[JOIN_TILE];
  __shared__ uint32_t s_run_elements[4][256];
  __shared__ uint32_t s_run_pairs[4][256];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t partitions = 1u << a.radix_bits;
  for (uint32_t w = 0; w < 4; ++w) { s_run_elements[w][tid] = 0; s_run_pairs[w][tid] = 0; }
  __syncthreads();
  uint32_t chunk, row_begin, row_count;
  tile_rows(a, blockIdx.x, &chunk, &row_begin, &row_count);

  // (a) evaluate every row once: wave w owns rows [w*512, (w+1)*512), round = 64 consecutive rows
  for (uint32_t round = 0; round < 8; ++round) {
    const uint32_t r = wave * 512 + round * 64 + lane;
    ProbeRow p{INVALID_PARTITION, 0, 0, false};
    if (r < row_count) p = probe_row(a, chunk, row_begin + r);
    s_meta[r] = static_cast<uint16_t>(p.partition | (p.null_partner ? 0x8000u : 0u));
    s_emit[r] = p.emit;
    s_start[r] = p.start;
    if (p.partition != INVALID_PARTITION) {
      atomicAdd(&s_run_elements[wave][p.partition], 1u);
      if (p.emit) atomicAdd(&s_run_pairs[wave][p.partition], p.emit);
    }
  }
  __syncthreads();
  // (b) thread = partition: exclusive prefix over the waves
  if (tid < partitions) {
    uint32_t run_e = 0, run_p = 0;
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t e = s_run_elements[w][tid], p = s_run_pairs[w][tid];
      s_run_elements[w][tid] = run_e;
      s_run_pairs[w][tid] = run_p;
      run_e += e;
      run_p += p;
    }
  }
  __syncthreads();
  // (c) stable ranking inside the wave, round by round
  const hy_row_id null_row{0xFFFFFFFFu, 0xFFFFFFFFu};
  for (uint32_t round = 0; round < 8; ++round) {
    const uint32_t round_base = wave * 512 + round * 64;
    const uint32_t r = round_base + lane;
    const uint32_t meta = s_meta[r];
    const uint32_t partition = meta & 0x1FF;
    const bool valid = partition != INVALID_PARTITION;
    const uint64_t peers = match_any(partition, valid, a.radix_bits);
    uint32_t pairs_before = 0, pairs_total = 0;
    if (valid) {
      uint64_t lower = peers & ((1ull << lane) - 1);
      while (lower) {
        const uint32_t j = __ffsll(static_cast<long long>(lower)) - 1;
        lower &= lower - 1;
        pairs_before += s_emit[round_base + j];
      }
      const uint32_t emit = s_emit[r];
      pairs_total = pairs_before + emit;
      const uint32_t element_rank = s_run_elements[wave][partition] + __popcll(peers & ((1ull << lane) - 1));
      const uint32_t pair_rank = s_run_pairs[wave][partition] + pairs_before;
      const size_t cell = static_cast<size_t>(partition) * a.n_tiles + blockIdx.x;
      const uint64_t pair_pos = a.base_pairs[cell] + pair_rank;
      // 131 070-element cuts (join_hash_steps.hpp:655-660)
      const uint64_t origin = a.radix_bits ? a.partition_element_origin[partition] : a.partition_element_origin[chunk];
      const uint64_t element_in_partition = a.base_elements[cell] + element_rank - origin;
      if (element_in_partition % PROBE_SIZE_PER_CHUNK == 0) {
        const uint32_t slice_base = a.radix_bits ? a.partition_slice_base[partition] : a.partition_slice_base[chunk];
        a.slice_offsets[slice_base + element_in_partition / PROBE_SIZE_PER_CHUNK] = pair_pos;
      }
      if (emit) {
        const hy_row_id probe_id{chunk, row_begin + r};
        const bool null_partner = meta & 0x8000u;
        const uint32_t start = s_start[r];
        for (uint32_t t = 0; t < emit; ++t) {
          a.probe_out[pair_pos + t] = probe_id;
          if (a.build_out) a.build_out[pair_pos + t] = null_partner ? null_row : a.dir.row_ids[start + t];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers >> lane) >> 1 == 0) {   // highest peer advances the running counters of its partition
      s_run_elements[wave][partition] += static_cast<uint32_t>(__popcll(peers));
      s_run_pairs[wave][partition] += pairs_total;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------

static uint32_t calculate_radix_bits(uint64_t build_rows) {   // join_hash.cpp:70-114
  const double l2_cache_max_usable = 1024000 * 0.75;
  const double complete_hash_map_size = static_cast<double>(build_rows) * static_cast<double>(sizeof(uint32_t)) / 0.8;
  const double cluster_count = std::max(1.0, complete_hash_map_size / l2_cache_max_usable);
  return static_cast<uint32_t>(std::min<size_t>(8, static_cast<size_t>(std::ceil(std::log2(cluster_count)))));
}

struct DeviceBuffer;
static hy_status exclusive_scan(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t stream);

// Temporary device buffers come from a per-thread pool of power-of-two blocks that is reused across calls
// (hipMalloc/hipFree cost ~100 us each and synchronise the device).
struct BufferPool {
  std::vector<std::pair<size_t, void*>> free_blocks;
  ~BufferPool() { for (auto& b : free_blocks) (void)hipFree(b.second); }
};
static thread_local BufferPool t_pool;

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t capacity = 0;
  hy_status alloc(size_t bytes) {
    size_t rounded = 4096;
    while (rounded < bytes) rounded <<= 1;
    for (size_t i = 0; i < t_pool.free_blocks.size(); ++i) {
      if (t_pool.free_blocks[i].first == rounded) {
        ptr = t_pool.free_blocks[i].second;
        capacity = rounded;
        t_pool.free_blocks.erase(t_pool.free_blocks.begin() + i);
        return HY_OK;
      }
    }
    hipError_t err = hipMalloc(&ptr, rounded);
    if (err != hipSuccess) {   // release the pool and retry once
      for (auto& b : t_pool.free_blocks) (void)hipFree(b.second);
      t_pool.free_blocks.clear();
      err = hipMalloc(&ptr, rounded);
    }
    if (err != hipSuccess) { ptr = nullptr; return fail(HY_ERR_DEVICE, "hipMalloc(%zu) failed: %s", rounded, hipGetErrorString(err)); }
    capacity = rounded;
    return HY_OK;
  }
  ~DeviceBuffer() { if (ptr) t_pool.free_blocks.emplace_back(capacity, ptr); }
  template <typename T> T* as() const { return static_cast<T*>(ptr); }
};

// out[0..n) = exclusive prefix sums of in, out[n] = total
static hy_status exclusive_scan(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t stream) {
  const uint32_t n_blocks = static_cast<uint32_t>((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  if (n_blocks == 0) {
    HY_HIP(hipMemsetAsync(out, 0, 8, stream));
    return HY_OK;
  }
  DeviceBuffer sums;
  HY_TRY(sums.alloc(8 * size_t{n_blocks}));
  hipLaunchKernelGGL(scan_block_sums, dim3(n_blocks), dim3(256), 0, stream, in, n, sums.as<uint64_t>());
  hipLaunchKernelGGL(scan_block_offsets, dim3(1), dim3(1024), 0, stream, sums.as<uint64_t>(), n_blocks, out + n);
  hipLaunchKernelGGL(scan_blocks, dim3(n_blocks), dim3(256), 0, stream, in, n, sums.as<uint64_t>(), out);
  HY_HIP(hipStreamSynchronize(stream));   // `sums` is freed on return
  return HY_OK;
}

static bool is_integer_column(const hy_column* c) { return c->data_type == HY_TYPE_INT || c->data_type == HY_TYPE_LONG; }

struct BuildSide {
  DeviceBuffer keys, rows, keys_tmp, rows_tmp, dir, bloom, flags;
  uint64_t n = 0;
  Directory directory{};
  bool any_null = false;
};

// Materialise + (sort) + directory.  `bloom_in` (device) filters the build side (no observable effect, kept for the
// reference's element counts); `bloom_out` receives the build side's filter if wanted.
static hy_status prepare_build(const hy_column* build, bool keep_nulls, bool want_bloom, BuildSide& b, hipStream_t stream) {
  const uint32_t n_slices = build->n_slices;
  DeviceBuffer counts, offsets;
  HY_TRY(counts.alloc(4 * size_t{n_slices + 1}));
  HY_TRY(offsets.alloc(8 * size_t{n_slices + 2}));
  HY_TRY(b.flags.alloc(64));
  HY_HIP(hipMemsetAsync(b.flags.ptr, 0, 64, stream));
  if (want_bloom) {
    HY_TRY(b.bloom.alloc(BLOOM_BITS));
    HY_HIP(hipMemsetAsync(b.bloom.ptr, 0, BLOOM_BITS, stream));
  }
  MaterializeArgs m{};
  m.segments = build->d_segments;
  m.slices = build->d_slices;
  m.n_slices = n_slices;
  m.keep_nulls = keep_nulls;
  m.bloom_in = nullptr;
  m.bloom_out = want_bloom ? b.bloom.as<uint8_t>() : nullptr;
  m.slice_counts = counts.as<uint32_t>();
  m.any_null = b.flags.as<uint32_t>() + 2;
  uint64_t total = 0;
  if (n_slices) {
    hipLaunchKernelGGL(join_materialize<0>, dim3(n_slices), dim3(256), 0, stream, m);
    hipLaunchKernelGGL(scan_counts, dim3(1), dim3(1024), 0, stream, counts.as<uint32_t>(), offsets.as<uint64_t>(), n_slices);
    HY_HIP(hipMemcpyAsync(&total, offsets.as<uint64_t>() + n_slices, 8, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
  }
  b.n = total;
  HY_TRY(b.keys.alloc(8 * total));
  HY_TRY(b.rows.alloc(8 * total));
  if (total) {
    m.slice_offsets = offsets.as<uint64_t>();
    m.keys = b.keys.as<uint64_t>();
    m.row_ids = b.rows.as<hy_row_id>();
    hipLaunchKernelGGL(join_materialize<1>, dim3(n_slices), dim3(256), 0, stream, m);
    uint32_t* unsorted = b.flags.as<uint32_t>();
    unsigned long long* key_or = reinterpret_cast<unsigned long long*>(b.flags.as<uint32_t>() + 4);
    hipLaunchKernelGGL(check_sorted, dim3(static_cast<uint32_t>((total + 1023) / 1024)), dim3(1024), 0, stream, b.keys.as<uint64_t>(), total, unsorted, key_or);
    uint32_t host_flags[8];
    HY_HIP(hipMemcpyAsync(host_flags, b.flags.ptr, 32, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
    b.any_null = host_flags[2] != 0;
    if (host_flags[0]) {   // not sorted: stable LSD radix sort, only over the bytes that are not constant zero
      uint64_t key_bits;
      std::memcpy(&key_bits, &host_flags[4], 8);
      HY_TRY(b.keys_tmp.alloc(8 * total));
      HY_TRY(b.rows_tmp.alloc(8 * total));
      const uint32_t n_tiles = static_cast<uint32_t>((total + SORT_TILE - 1) / SORT_TILE);
      DeviceBuffer hist, bases;
      HY_TRY(hist.alloc(4 * size_t{256} * n_tiles));
      HY_TRY(bases.alloc(8 * (size_t{256} * n_tiles + 1)));
      uint64_t* src_keys = b.keys.as<uint64_t>();
      hy_row_id* src_rows = b.rows.as<hy_row_id>();
      uint64_t* dst_keys = b.keys_tmp.as<uint64_t>();
      hy_row_id* dst_rows = b.rows_tmp.as<hy_row_id>();
      for (uint32_t shift = 0; shift < 64; shift += 8) {
        if (((key_bits >> shift) & 0xFF) == 0 && !(key_bits >> 63)) continue;   // byte is zero in every key
        hipLaunchKernelGGL(sort_histogram, dim3(n_tiles), dim3(256), 0, stream, src_keys, total, shift, hist.as<uint32_t>(), n_tiles);
        HY_TRY(exclusive_scan(hist.as<uint32_t>(), bases.as<uint64_t>(), uint64_t{256} * n_tiles, stream));
        hipLaunchKernelGGL(sort_scatter, dim3(n_tiles), dim3(256), 0, stream, src_keys, src_rows, dst_keys, dst_rows, total, shift, bases.as<uint64_t>(), n_tiles);
        std::swap(src_keys, dst_keys);
        std::swap(src_rows, dst_rows);
      }
      if (src_keys != b.keys.as<uint64_t>()) {
        std::swap(b.keys.ptr, b.keys_tmp.ptr);
        std::swap(b.keys.capacity, b.keys_tmp.capacity);
        std::swap(b.rows.ptr, b.rows_tmp.ptr);
        std::swap(b.rows.capacity, b.rows_tmp.capacity);
      }
    }
  }
  // directory
  Directory& d = b.directory;
  d.keys = b.keys.as<uint64_t>();
  d.row_ids = b.rows.as<hy_row_id>();
  d.n = total;
  d.key_min = d.key_max = 0;
  d.shift = 0;
  d.n_buckets = 1;
  if (total) {
    HY_HIP(hipMemcpyAsync(&d.key_min, d.keys, 8, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipMemcpyAsync(&d.key_max, d.keys + (total - 1), 8, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
    uint32_t buckets = 1;
    while (buckets < total / 8 && buckets < (1u << 26)) buckets <<= 1;   // ~8 keys per bucket on uniform keys
    const uint64_t range = d.key_max - d.key_min;
    uint32_t shift = 0;
    while (shift < 64 && (range >> shift) >= buckets) ++shift;
    d.shift = shift;
    d.n_buckets = static_cast<uint32_t>((range >> shift) + 1);
  }
  HY_TRY(b.dir.alloc(4 * (size_t{d.n_buckets} + 2)));
  d.dir = b.dir.as<uint32_t>();
  if (total) {
    hipLaunchKernelGGL(directory_fill, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, stream, d.keys, total, d.key_min, d.shift, d.n_buckets, b.dir.as<uint32_t>());
  } else {
    HY_HIP(hipMemsetAsync(b.dir.ptr, 0, 4 * (size_t{d.n_buckets} + 2), stream));
  }
  return HY_OK;
}

static hy_status run_join(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result, bool count_only, uint64_t* count_out) {
  if (mode == HY_JOIN_FULL_OUTER || mode == HY_JOIN_CROSS || mode > HY_JOIN_ANTI_NULL_AS_FALSE) return fail(HY_ERR_UNSUPPORTED, "JoinHash does not support join mode %u (join_hash.cpp:38-44)", mode);
  if (!is_integer_column(left) || !is_integer_column(right)) return fail(HY_ERR_UNSUPPORTED, "only int32/int64 join keys run on the device (std::hash of float/string keys is not pinned)");
  hipStream_t stream = current_stream();
  // side selection (join_hash.cpp:139-155)
  const bool build_right = mode == HY_JOIN_LEFT || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE || mode == HY_JOIN_SEMI ||
                           (mode == HY_JOIN_INNER && left->rows > right->rows);
  const hy_column* build = build_right ? right : left;
  const hy_column* probe = build_right ? left : right;
  const bool semi_anti = mode == HY_JOIN_SEMI || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE;
  const bool keep_nulls_build = mode == HY_JOIN_ANTI_NULL_AS_TRUE;   // join_hash.cpp:284-286
  const bool keep_nulls_probe = mode == HY_JOIN_LEFT || mode == HY_JOIN_RIGHT || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE;
  uint32_t radix_bits = calculate_radix_bits(build->rows);
  if (result && result->radix_bits != 0xFFFFFFFFu) radix_bits = result->radix_bits;
  if (radix_bits > 8) return fail(HY_ERR_INVALID, "radix_bits %u > 8", radix_bits);
  const bool host_result = !result || result->mem == HY_MEM_HOST;

  // build side; its Bloom filter is applied to the probe side only when the build side is materialised first
  const bool probe_filtered = build->rows < probe->rows && !keep_nulls_probe;   // join_hash.cpp:365-381
  BuildSide b;
  HY_TRY(prepare_build(build, keep_nulls_build, probe_filtered, b, stream));

  if (result) {
    result->radix_bits = radix_bits;
    result->left_is_build = build_right ? 0 : 1;
    result->n_slices = 0;
    result->n_pairs = 0;
  }
  if (mode == HY_JOIN_ANTI_NULL_AS_TRUE && b.any_null) {   // join_hash.cpp:483-494
    if (count_out) *count_out = 0;
    if (result && result->slice_offsets && host_result) result->slice_offsets[0] = 0;
    if (result && result->slice_offsets && !host_result) HY_HIP(hipMemsetAsync(result->slice_offsets, 0, 8, stream));
    return HY_OK;
  }

  // probe pass 1
  const uint32_t n_tiles = probe->n_slices * 4;
  const uint32_t partitions = 1u << radix_bits;
  const size_t cells = size_t{partitions} * n_tiles;
  DeviceBuffer hist_e, hist_p, base_e, base_p;
  HY_TRY(hist_e.alloc(4 * (cells + 1)));
  HY_TRY(hist_p.alloc(4 * (cells + 1)));
  HY_TRY(base_e.alloc(8 * (cells + 2)));
  HY_TRY(base_p.alloc(8 * (cells + 2)));
  ProbeArgs a{};
  a.segments = probe->d_segments;
  a.slices = probe->d_slices;
  a.n_tiles = n_tiles;
  a.mode = mode;
  a.radix_bits = radix_bits;
  a.keep_nulls = keep_nulls_probe;
  a.build_rows_zero = build->rows == 0;
  a.build_bloom = probe_filtered ? b.bloom.as<uint8_t>() : nullptr;
  a.dir = b.directory;
  a.hist_elements = hist_e.as<uint32_t>();
  a.hist_pairs = hist_p.as<uint32_t>();
  // Which scanned positions the host needs: the start of every partition (radix) or of every probe chunk
  // (radix_bits == 0: "partitions" are the probe chunks), plus the grand totals.
  const uint32_t n_groups = radix_bits ? partitions : probe->n_chunks;
  std::vector<uint64_t> group_first_cell(size_t{n_groups} + 1, cells);
  if (radix_bits) {
    for (uint32_t p = 0; p <= partitions; ++p) group_first_cell[p] = size_t{p} * n_tiles;
  } else {
    uint64_t tile = 0;
    for (uint32_t c = 0; c < probe->n_chunks; ++c) {
      const uint32_t chunk_slices = (probe->host_segments[c].size + SLICE_ROWS - 1) / SLICE_ROWS;
      group_first_cell[c] = tile;
      tile += (chunk_slices ? chunk_slices : 1) * 4;   // the 4 tiles of each of its slices, consecutive
    }
    group_first_cell[n_groups] = tile;
  }
  std::vector<uint64_t> group_origin(size_t{n_groups} + 1, 0);
  uint64_t n_pairs = 0;
  DeviceBuffer d_index, d_origin;
  HY_TRY(d_index.alloc(8 * (size_t{n_groups} + 1)));
  HY_TRY(d_origin.alloc(8 * (size_t{n_groups} + 1)));
  if (n_tiles) {
    hipLaunchKernelGGL(probe_histogram, dim3(n_tiles), dim3(256), 0, stream, a);
    HY_TRY(exclusive_scan(hist_e.as<uint32_t>(), base_e.as<uint64_t>(), uint64_t{cells}, stream));
    HY_TRY(exclusive_scan(hist_p.as<uint32_t>(), base_p.as<uint64_t>(), uint64_t{cells}, stream));
    HY_HIP(hipMemcpyAsync(d_index.ptr, group_first_cell.data(), 8 * (size_t{n_groups} + 1), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gather_u64, dim3((n_groups + 256) / 256), dim3(256), 0, stream, base_e.as<uint64_t>(), d_index.as<uint64_t>(), d_origin.as<uint64_t>(), n_groups + 1);
    HY_HIP(hipMemcpyAsync(group_origin.data(), d_origin.ptr, 8 * (size_t{n_groups} + 1), hipMemcpyDeviceToHost, stream));
    HY_HIP(hipMemcpyAsync(&n_pairs, base_p.as<uint64_t>() + cells, 8, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
  }
  if (count_out) *count_out = n_pairs;
  if (count_only) return HY_OK;

  // slices: every 131 070 materialised elements of a group start a new output PosList
  std::vector<uint32_t> slice_base(n_groups ? n_groups : 1, 0);
  uint64_t n_slices = 0;
  for (uint32_t g = 0; g < n_groups; ++g) {
    slice_base[g] = static_cast<uint32_t>(n_slices);
    n_slices += (group_origin[g + 1] - group_origin[g] + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK;
  }
  result->n_slices = static_cast<uint32_t>(n_slices);
  result->n_pairs = n_pairs;
  if (n_slices > result->slice_capacity) return fail(HY_ERR_CAPACITY, "join produces %llu output PosLists, slice capacity is %u", static_cast<unsigned long long>(n_slices), result->slice_capacity);
  if (n_pairs > result->capacity) return fail(HY_ERR_CAPACITY, "join produces %llu pairs, capacity is %llu", static_cast<unsigned long long>(n_pairs), static_cast<unsigned long long>(result->capacity));

  hy_row_id* user_build = result->left_is_build ? result->left_pos : result->right_pos;
  hy_row_id* user_probe = result->left_is_build ? result->right_pos : result->left_pos;
  if (!user_probe && n_pairs) return fail(HY_ERR_INVALID, "join result: PosList buffer for the probe side missing");
  if (!semi_anti && !user_build && n_pairs) return fail(HY_ERR_INVALID, "join result: PosList buffer for the build side missing");
  if (!result->slice_offsets) return fail(HY_ERR_INVALID, "join result: slice_offsets missing");

  DeviceBuffer d_slice_base, d_build_out, d_probe_out, d_slice_offsets;
  HY_TRY(d_slice_base.alloc(4 * slice_base.size()));
  HY_HIP(hipMemcpyAsync(d_slice_base.ptr, slice_base.data(), 4 * slice_base.size(), hipMemcpyHostToDevice, stream));
  hy_row_id* dev_build = user_build;
  hy_row_id* dev_probe = user_probe;
  uint64_t* dev_slice_offsets = result->slice_offsets;
  if (host_result) {
    if (!semi_anti) { HY_TRY(d_build_out.alloc(8 * n_pairs)); dev_build = d_build_out.as<hy_row_id>(); }
    HY_TRY(d_probe_out.alloc(8 * n_pairs));
    dev_probe = d_probe_out.as<hy_row_id>();
    HY_TRY(d_slice_offsets.alloc(8 * (n_slices + 2)));
    dev_slice_offsets = d_slice_offsets.as<uint64_t>();
  }
  a.base_elements = base_e.as<uint64_t>();
  a.base_pairs = base_p.as<uint64_t>();
  a.partition_element_origin = d_origin.as<uint64_t>();
  a.partition_slice_base = d_slice_base.as<uint32_t>();
  a.build_out = semi_anti ? nullptr : dev_build;
  a.probe_out = dev_probe;
  a.slice_offsets = dev_slice_offsets;
  if (n_tiles) {
    profile_begin(stream);
    hipLaunchKernelGGL(probe_scatter, dim3(n_tiles), dim3(256), 0, stream, a);
    profile_end(stream);
  }
  HY_HIP(hipMemcpyAsync(dev_slice_offsets + n_slices, &result->n_pairs, 8, hipMemcpyHostToDevice, stream));
  HY_HIP(hipGetLastError());
  if (host_result) {
    if (n_pairs) {
      HY_HIP(hipMemcpyAsync(user_probe, dev_probe, 8 * n_pairs, hipMemcpyDeviceToHost, stream));
      if (!semi_anti) HY_HIP(hipMemcpyAsync(user_build, dev_build, 8 * n_pairs, hipMemcpyDeviceToHost, stream));
    }
    HY_HIP(hipMemcpyAsync(result->slice_offsets, dev_slice_offsets, 8 * (n_slices + 1), hipMemcpyDeviceToHost, stream));
  }
  HY_HIP(hipStreamSynchronize(stream));   // the temporaries above are freed on return
  return HY_OK;
}

}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_join_hash(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result) {
  if (!left || !right || !result) return fail(HY_ERR_INVALID, "hy_join_hash: null argument");
  return run_join(left, right, mode, result, false, nullptr);
}

hy_status hy_join_hash_count(const hy_column* left, const hy_column* right, uint32_t mode, uint64_t* n_pairs) {
  if (!left || !right || !n_pairs) return fail(HY_ERR_INVALID, "hy_join_hash_count: null argument");
  return run_join(left, right, mode, nullptr, true, n_pairs);
}

hy_status hy_join_hash_radix_bits(uint64_t build_rows, uint64_t probe_rows, uint32_t* radix_bits) {
  (void)probe_rows;
  if (!radix_bits) return fail(HY_ERR_INVALID, "hy_join_hash_radix_bits: null argument");
  *radix_bits = calculate_radix_bits(build_rows);
  return HY_OK;
}

}  // extern "C"
