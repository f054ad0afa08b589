// join.hip -- JoinHash on MI355X: equi-joins of numeric columns (int32 / int64 / float / double, any two), all JoinHash modes,
// up to four secondary predicates.
//
// What it replaces (reference, CPU):
//   JoinHash::_on_execute / JoinHashImpl::_on_execute                 operators/join_hash.cpp:116-225, 270-572
//   materialize_input / partition_by_radix / build / probe(_semi_anti) operators/join_hash/join_hash_steps.hpp:274-922
//
// The reference's output order is fully determined by its algorithm: pairs come radix partition by radix partition
// (low `radix_bits` bits of the key's std::hash; `radix_bits` == 0: probe chunk by probe chunk), inside a partition by probe
// row, inside a probe row by build row (hash-table insertion order == (chunk, row) order), and every 131 070 materialised
// probe elements of a partition start a new output PosList.  The device code produces exactly that order without
// materialising the partitions themselves.  Two build-side structures, chosen per join (prepare_build):
//   rank table   unique integer build keys whose range is at most ~64 x their number (a primary key, dense or dbgen-sparse):
//                one 8-byte entry {32 presence bits, rank of the word's first key} per 32 key values; a lookup is ONE dependent
//                load, the partner's rank = base + popcount(bits below the key).  A dense, sorted build column of equal-sized
//                chunks is read in place (dense_key_stats, rank_table_fill_dense: no key or RowID arrays at all) and the rank IS
//                the row number; unsorted unique keys mark their bits with atomicOr (a bit that was already set = a
//                duplicate -> fall back), block sums turn into bases, and a rank -> row array is scattered.
//   directory    everything else (duplicate keys, sparse keys, float / double keys, secondary predicates): (key, RowID) in row
//                order, an 8-bit LSD radix sort by key only if the keys are not already sorted, and an order-preserving
//                bucket directory (key - min) >> shift over the sorted keys; equal keys are adjacent in build-row order, a
//                probe hit is (start, count).
// Probe side, two passes over the (still encoded) probe column in 4096-row tiles:
//   1. count     per (tile, partition): materialised probe elements and output pairs (rt_stream_count: one wave per tile,
//                16-byte loads; rt_probe_count / probe_count: one workgroup per tile for the general decoders)
//      scans + plan_output (capacity check, PosList plan -- on the device) + probe_cuts (the 131 070-element cuts)
//   2. emit      every (partition, tile) cell has its output range; rt_probe_emit re-evaluates the tile's rows (nothing is
//                handed over from pass 1), ranks the pairs of a partition inside the tile (one returning LDS atomic per
//                match-any group of a wave), stages them partition by partition in LDS and writes contiguous runs with
//                16-byte nontemporal stores.  Tiles with multi-partner rows take probe_emit_generic.
// HBM traffic at config 3: build keys (twice: statistics, fill) + 2 x probe keys + 16 B per pair + the rank table; no
// 12-byte PartitionedElement arrays are ever written.  The Bloom filters of the reference are reproduced only where they are
// observable: the build side's filter decides which probe elements count as "materialised" (it shifts the cuts).
#include "hy_device.hpp"
#include "hy_decode.hpp"
#include "hy_arithmetic.hpp"
#include "hy_scan_job.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstring>
#include <type_traits>

namespace hy {

constexpr uint32_t JOIN_TILE = 4096;                 // probe rows per workgroup tile: half a slice
constexpr uint32_t JOIN_THREADS = 512;
constexpr uint32_t JOIN_WAVES = JOIN_THREADS / 64;
constexpr uint32_t JOIN_WAVE_ROWS = JOIN_TILE / JOIN_WAVES;   // consecutive rows of a wave
constexpr uint32_t JOIN_ROUNDS = JOIN_WAVE_ROWS / 64;
constexpr uint32_t JOIN_STAGE = JOIN_TILE + 256;     // pairs a tile can stage in LDS before it falls back to direct writes
constexpr uint32_t MAX_PARTITIONS = 256;             // radix_bits <= 8
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
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

// ---- float / double keys, also joined with integer columns -----------------------------------------------------------------
// JoinHashTraits (join_hash_traits.hpp:15-40): both sides are cast to one HashedType -- the larger floating type, or THE
// floating type of an integer x floating join -- hashed with std::hash<HashedType> and compared there.  Such a join keeps
// the BIT PATTERN of the HashedType value as its 64-bit key (float: zero-extended; -0.0 as +0.0, they compare equal and
// both hash to 0): equal bits <=> equal keys, except NaN, which finds nothing.  The radix partition and the Bloom filter
// take std::hash of the value, which libstdc++ defines as 0 for zero and _Hash_bytes (libsupc++/hash_bytes.cc, seed
// 0xc70f6907) over the 4 / 8 bytes otherwise -- restated here, pinned in tests/test_oracle_join.py.
// hashed_type: 0 (integer keys, std::hash is the identity) | HY_TYPE_FLOAT | HY_TYPE_DOUBLE.
__device__ __forceinline__ uint64_t join_hash(uint64_t key, uint32_t hashed_type) {
  if (hashed_type == 0 || key == 0) return key;
  constexpr uint64_t mul = (0xc6a4a793ull << 32) + 0x5bd1e995ull;
  auto shift_mix = [](uint64_t v) { return v ^ (v >> 47); };
  uint64_t hash;
  if (hashed_type == HY_TYPE_FLOAT) {
    hash = 0xc70f6907ull ^ (4 * mul);
    hash ^= key;   // the four bytes, little endian
    hash *= mul;
  } else {
    hash = 0xc70f6907ull ^ (8 * mul);
    hash ^= shift_mix(key * mul) * mul;
    hash *= mul;
  }
  hash = shift_mix(hash) * mul;
  return shift_mix(hash);
}

__device__ __forceinline__ bool key_is_nan(uint64_t key, uint32_t hashed_type) {
  if (hashed_type == HY_TYPE_FLOAT) return (static_cast<uint32_t>(key) & 0x7FFFFFFFu) > 0x7F800000u;
  if (hashed_type == HY_TYPE_DOUBLE) return (key & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull;
  return false;
}

// static_cast<HashedType>(value) of a row of any numeric column (data or reference segments), as the key above; true if NULL.
__device__ __forceinline__ bool hashed_key(const DevSegment* segments, uint32_t chunk, uint32_t row, uint32_t hashed_type, int64_t* key) {
  *key = 0;
  const Value v = column_value(segments, chunk, row);
  if (v.is_null) return true;
  const uint32_t type = segments[chunk].data_type;
  const bool is_float = type == HY_TYPE_FLOAT || type == HY_TYPE_DOUBLE;
  if (hashed_type == HY_TYPE_FLOAT) {
    const float f = is_float ? static_cast<float>(v.f) : static_cast<float>(v.i);
    *key = f == 0.0f ? 0 : static_cast<int64_t>(__float_as_uint(f));
  } else {
    const double d = is_float ? v.f : static_cast<double>(v.i);
    *key = d == 0.0 ? 0 : __double_as_longlong(d);
  }
  return false;
}

// The build side keeps its keys and RowIDs as narrow as the join allows: int32 columns as 32-bit keys (their unsigned
// order is the unsigned order of the sign-extended 64-bit keys), RowIDs packed into 32 bits when every chunk id and chunk
// offset of the build table is below 2^16 (Hyrise's default chunk size: always) -- half the bytes to write, sort and read.
template <bool KEY32> struct BuildKey { using type = uint64_t; };
template <> struct BuildKey<true> { using type = uint32_t; };
template <bool ID32> struct BuildRow { using type = hy_row_id; };
template <> struct BuildRow<true> { using type = uint32_t; };
template <bool ID32> __device__ __forceinline__ typename BuildRow<ID32>::type make_build_row(uint32_t chunk, uint32_t offset) {
  if constexpr (ID32) return chunk << 16 | offset;
  else return hy_row_id{chunk, offset};
}

// the 64-bit (sign-extended) key of a stored build key
__device__ __forceinline__ uint64_t key_bits_of(uint32_t key) { return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(key))); }
__device__ __forceinline__ uint64_t key_bits_of(uint64_t key) { return key; }

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
  void* keys;                     // MODE 1: uint32_t (int32 columns: KEY32) or uint64_t sign-extended key bits
  void* row_ids;                  // MODE 1: uint32_t chunk_id << 16 | chunk_offset (ID32) or hy_row_id
  uint32_t* any_null;             // set to 1 if a NULL was materialised (AntiNullAsTrue early-out)
  const uint64_t* row_base;       // dense: [n_chunks + 1] first row of every chunk
  uint32_t hashed_type;           // 0: integer keys | HY_TYPE_FLOAT | HY_TYPE_DOUBLE (see join_hash)
  const SliceView* views;         // the same slices as one record each (rank_table_fill_checked)
};

template <int MODE, bool KEY32, bool ID32>
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
      const bool is_null = a.hashed_type ? hashed_key(a.segments, slice.chunk, slice.row_begin + r, a.hashed_type, &key) : column_key(a.segments, slice.chunk, slice.row_begin + r, &key);
      keep = !is_null || a.keep_nulls;
      if (keep && !is_null && a.bloom_in && !a.keep_nulls) keep = bloom_test(a.bloom_in, join_hash(static_cast<uint64_t>(key), a.hashed_type));
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
      const bool is_null = a.hashed_type ? hashed_key(a.segments, slice.chunk, slice.row_begin + r, a.hashed_type, &key) : column_key(a.segments, slice.chunk, slice.row_begin + r, &key);
      const uint64_t pos = base + s_offset[k][wave] + __popcll(ballot & ((1ull << lane) - 1));
      static_cast<typename BuildKey<KEY32>::type*>(a.keys)[pos] = static_cast<typename BuildKey<KEY32>::type>(key);
      static_cast<typename BuildRow<ID32>::type*>(a.row_ids)[pos] = make_build_row<ID32>(slice.chunk, slice.row_begin + r);
      if (is_null && a.any_null) *a.any_null = 1;
      if (a.bloom_out) a.bloom_out[static_cast<uint32_t>(join_hash(static_cast<uint64_t>(key), a.hashed_type)) & (BLOOM_BITS - 1)] = 1;
    }
  }
}

// Build columns that cannot hold NULLs (value / FrameOfReference segments without a null vector): every row is
// materialised at its own row number -- no counting, no ballots, eight rows of a lane in flight at a time.
template <bool KEY32, bool ID32>
__global__ __launch_bounds__(256) void join_materialize_dense(MaterializeArgs a) {
  const uint32_t tid = threadIdx.x;
  const Slice slice = a.slices[blockIdx.x];
  const DevSegment s = a.segments[slice.chunk];
  const uint64_t base = a.row_base[slice.chunk] + slice.row_begin;   // every row is materialised: offsets are row numbers
  constexpr uint32_t BATCH = 8;
#pragma unroll 1
  for (uint32_t block = 0; block < SLICE_ROWS / 256 / BATCH; ++block) {
    uint32_t r[BATCH];
    int64_t key[BATCH];
#pragma unroll
    for (uint32_t i = 0; i < BATCH; ++i) {
      r[i] = (block * BATCH + i) * 256 + tid;
      const uint32_t row = slice.row_begin + (r[i] < slice.row_count ? r[i] : 0);
      if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
        key[i] = static_cast<int32_t>(jload_compressed(s.data, s.width, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]));
      } else if (s.data_type == HY_TYPE_INT) {
        key[i] = static_cast<const int32_t*>(s.data)[row];
      } else {
        key[i] = static_cast<const int64_t*>(s.data)[row];
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < BATCH; ++i) {
      if (r[i] >= slice.row_count) continue;
      static_cast<typename BuildKey<KEY32>::type*>(a.keys)[base + r[i]] = static_cast<typename BuildKey<KEY32>::type>(key[i]);
      static_cast<typename BuildRow<ID32>::type*>(a.row_ids)[base + r[i]] = make_build_row<ID32>(slice.chunk, slice.row_begin + r[i]);
      if (a.bloom_out) a.bloom_out[static_cast<uint32_t>(key[i]) & (BLOOM_BITS - 1)] = 1;
    }
  }
}

// ---- dense build columns read in place ------------------------------------------------------------------------------------
// A dense build column (above) that turns out to be sorted and duplicate-free -- a primary key -- needs neither a key array
// nor a RowID array: the rank table is filled from the column itself and a key's rank is its row number.  Two passes over
// the column: the statistics that decide (order, equal neighbours, smallest / largest key: the flags of check_sorted), then
// the table.  Rows are visited as row = k * 256 + tid: consecutive lanes hold consecutive rows.
__device__ __forceinline__ int64_t dense_key(const DevSegment& s, uint32_t row) {
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) return static_cast<int32_t>(jload_compressed(s.data, s.width, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]));
  if (s.data_type == HY_TYPE_INT) return static_cast<const int32_t*>(s.data)[row];
  return static_cast<const int64_t*>(s.data)[row];
}

// The key that follows the slice's last row (the first row of the next slice that has rows), or `fallback`.
__device__ __forceinline__ int64_t key_after_slice(const MaterializeArgs& a, uint32_t slice_index, int64_t fallback, bool* exists) {
  *exists = false;
  for (uint32_t next = slice_index + 1; next < a.n_slices; ++next) {
    const Slice after = a.slices[next];
    if (after.row_count == 0) continue;
    *exists = true;
    return dense_key(a.segments[after.chunk], after.row_begin);
  }
  return fallback;
}

__global__ __launch_bounds__(256) void dense_key_stats(MaterializeArgs a, uint64_t* partials) {   // partials: [n_slices][4] OR | min ^ sign | max ^ sign | flags
  __shared__ uint64_t s_bits[4], s_min[4], s_max[4];
  __shared__ uint32_t s_flags[3];
  const uint32_t tid = threadIdx.x;
  const Slice slice = a.slices[blockIdx.x];
  if (slice.row_count == 0) {
    if (tid == 0) { uint64_t* record = partials + 4 * size_t{blockIdx.x}; record[0] = 0; record[1] = ~0ull; record[2] = 0; record[3] = 0; }
    return;
  }
  if (tid < 3) s_flags[tid] = 0;
  __syncthreads();
  const DevSegment s = a.segments[slice.chunk];
  constexpr uint64_t SIGN = 1ull << 63;
  uint64_t bits = 0, lowest = ~0ull, highest = 0;
  bool unsorted = false, equal = false, unsorted_signed = false;
  if (s.encoding == HY_ENC_UNENCODED && s.data_type == HY_TYPE_INT && !(s.flags & SEG_UNALIGNED)) {
    // int32 values: four consecutive rows per lane and load (16 bytes), all eight loads of the slice in flight, plus the one
    // value behind each group; everything in 32-bit arithmetic (signed order; the unsigned order of the sign-extended keys
    // is the unsigned order of the 32-bit patterns)
    const int32_t* values = static_cast<const int32_t*>(s.data) + slice.row_begin;
    constexpr uint32_t GROUPS = SLICE_ROWS / 1024;
    u32x4_t group[GROUPS];
    int32_t after[GROUPS];
#pragma unroll
    for (uint32_t i = 0; i < GROUPS; ++i) {
      const uint32_t first = i * 1024 + tid * 4;
      group[i] = *reinterpret_cast<const u32x4_t*>(values + (first < slice.row_count ? first : 0));
      after[i] = values[first + 4 < slice.row_count ? first + 4 : 0];
    }
    int32_t low32 = 0x7FFFFFFF, high32 = static_cast<int32_t>(0x80000000u);
    uint32_t or32 = 0;
#pragma unroll
    for (uint32_t i = 0; i < GROUPS; ++i) {
      const uint32_t first = i * 1024 + tid * 4;
      const int32_t k[5] = {static_cast<int32_t>(group[i].x), static_cast<int32_t>(group[i].y), static_cast<int32_t>(group[i].z), static_cast<int32_t>(group[i].w), after[i]};
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        if (first + e >= slice.row_count) continue;
        or32 |= static_cast<uint32_t>(k[e]);
        low32 = k[e] < low32 ? k[e] : low32;
        high32 = k[e] > high32 ? k[e] : high32;
        bool has_next = first + e + 1 < slice.row_count;
        int32_t next = k[e + 1];
        if (!has_next) next = static_cast<int32_t>(key_after_slice(a, blockIdx.x, 0, &has_next));   // (one thread per slice)
        if (has_next) {
          if (static_cast<uint32_t>(k[e]) > static_cast<uint32_t>(next)) unsorted = true;
          if (k[e] == next) equal = true;
          if (k[e] > next) unsorted_signed = true;
        }
      }
    }
    if (low32 <= high32) {
      bits = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(or32)));   // (sign-extended like the keys: only its low half is looked at for 32-bit keys)
      lowest = static_cast<uint64_t>(static_cast<int64_t>(low32)) ^ SIGN;
      highest = static_cast<uint64_t>(static_cast<int64_t>(high32)) ^ SIGN;
    }
  } else {
    constexpr uint32_t BATCH = 8;
#pragma unroll 1
    for (uint32_t block = 0; block < SLICE_ROWS / 256 / BATCH; ++block) {
      if (block * BATCH * 256 >= slice.row_count) break;
      int64_t key[BATCH], next[BATCH];
#pragma unroll
      for (uint32_t i = 0; i < BATCH; ++i) {
        const uint32_t r = (block * BATCH + i) * 256 + tid;
        const uint32_t row = slice.row_begin + (r < slice.row_count ? r : 0);
        const uint32_t row_after = slice.row_begin + (r + 1 < slice.row_count ? r + 1 : 0);
        key[i] = dense_key(s, row);
        next[i] = dense_key(s, row_after);
      }
#pragma unroll
      for (uint32_t i = 0; i < BATCH; ++i) {
        const uint32_t r = (block * BATCH + i) * 256 + tid;
        if (r >= slice.row_count) continue;
        const uint64_t wide = static_cast<uint64_t>(key[i]);
        bits |= wide;
        lowest = (wide ^ SIGN) < lowest ? (wide ^ SIGN) : lowest;
        highest = (wide ^ SIGN) > highest ? (wide ^ SIGN) : highest;
        bool has_next = r + 1 < slice.row_count;
        int64_t after = next[i];
        if (!has_next) after = key_after_slice(a, blockIdx.x, 0, &has_next);   // (one thread per slice)
        if (has_next) {
          if (wide > static_cast<uint64_t>(after)) unsorted = true;
          if (key[i] == after) equal = true;
          if (key[i] > after) unsorted_signed = true;
        }
      }
    }
  }
  if (unsorted) s_flags[0] = 1;
  if (equal) s_flags[1] = 1;
  if (unsorted_signed) s_flags[2] = 1;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    bits |= __shfl_xor(bits, d, 64);
    const uint64_t other_low = __shfl_xor(lowest, d, 64), other_high = __shfl_xor(highest, d, 64);
    lowest = other_low < lowest ? other_low : lowest;
    highest = other_high > highest ? other_high : highest;
  }
  if ((tid & 63) == 0) { s_bits[tid >> 6] = bits; s_min[tid >> 6] = lowest; s_max[tid >> 6] = highest; }
  __syncthreads();
  if (tid == 0) {   // one record per slice (thousands of atomics on three words would take longer than the column: ~88 per microsecond)
    uint64_t all = 0, low = ~0ull, high = 0;
    for (uint32_t w = 0; w < 4; ++w) { all |= s_bits[w]; low = s_min[w] < low ? s_min[w] : low; high = s_max[w] > high ? s_max[w] : high; }
    uint64_t* record = partials + 4 * size_t{blockIdx.x};
    record[0] = all;
    record[1] = low;
    record[2] = high;
    record[3] = (s_flags[0] ? 1u : 0u) | (s_flags[1] ? 2u : 0u) | (s_flags[2] ? 4u : 0u);
  }
}

// The rank table of a sorted, duplicate-free dense column, from the column.  A slice's keys are consecutive keys of the
// column, so they fill a contiguous run of table words: the workgroup assembles the run in LDS (ds_or per key; the first
// key of a word also notes its row number -- the entry's base) and stores it with coalesced 8-byte writes.  Only the
// run's first and last word can hold keys of a neighbouring slice: those two go to the (zeroed) table with an atomic OR.
// A slice whose keys span more words than the LDS holds (a sparse stretch) sets its bits with one global atomic per key.
constexpr uint32_t FILL_WORDS = 4096;
constexpr uint32_t CHECKED_FILL_TICKETS = 32;   // second-level arrival counters of rank_table_fill_checked, 128 bytes apart behind the first
constexpr uint32_t CHECKED_FILL_WORDS = 2048;   // rank_table_fill_checked: 16 KB of LDS, eight workgroups per CU (sparser slices set their bits in the table itself)
typedef uint32_t u32x2_entry_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void rank_table_fill_dense(MaterializeArgs a, uint64_t key_min, u32x2_entry_t* entries) {
  __shared__ uint32_t s_bits[FILL_WORDS], s_base[FILL_WORDS];
  const uint32_t tid = threadIdx.x;
  const Slice slice = a.slices[blockIdx.x];
  if (slice.row_count == 0) return;
  const DevSegment s = a.segments[slice.chunk];
  const uint64_t first_row = a.row_base[slice.chunk] + slice.row_begin;   // rank of the slice's first key
  const uint64_t first_word = (static_cast<uint64_t>(dense_key(s, slice.row_begin)) - key_min) >> 5;
  const uint64_t last_word = (static_cast<uint64_t>(dense_key(s, slice.row_begin + slice.row_count - 1)) - key_min) >> 5;
  const uint64_t span = last_word - first_word + 1;
  const bool staged = span <= FILL_WORDS;
  // the key in front of the slice (the last row of the nearest earlier slice with rows): its word
  uint64_t word_before = ~0ull;
  if (tid == 0) {
    for (uint32_t before = blockIdx.x; before-- > 0;) {
      const Slice earlier = a.slices[before];
      if (earlier.row_count == 0) continue;
      word_before = (static_cast<uint64_t>(dense_key(a.segments[earlier.chunk], earlier.row_begin + earlier.row_count - 1)) - key_min) >> 5;
      break;
    }
  }
  if (staged) {
    for (uint32_t i = tid; i < span; i += 256) { s_bits[i] = 0; s_base[i] = 0; }
    __syncthreads();
  }
  auto place = [&](int64_t key, uint64_t previous_word, uint32_t r) {   // row r of the slice
    const uint64_t rel = static_cast<uint64_t>(key) - key_min;
    const uint64_t word = rel >> 5;
    const uint32_t bit = 1u << (rel & 31);
    const bool leader = previous_word != word;   // no earlier row shares the word: its row number is the entry's base
    if (staged) {
      atomicOr(&s_bits[word - first_word], bit);
      if (leader) s_base[word - first_word] = static_cast<uint32_t>(first_row + r) + 1u;   // (+ 1: 0 = the word's first key is not in this slice)
    } else {
      atomicOr(reinterpret_cast<uint32_t*>(entries + word), bit);
      if (leader) reinterpret_cast<uint32_t*>(entries + word)[1] = static_cast<uint32_t>(first_row + r);
    }
    if (a.bloom_out) a.bloom_out[static_cast<uint32_t>(key) & (BLOOM_BITS - 1)] = 1;
  };
  if (s.encoding == HY_ENC_UNENCODED && s.data_type == HY_TYPE_INT && !(s.flags & SEG_UNALIGNED)) {
    // int32 values: four consecutive rows per lane and load, all eight loads in flight, plus the value in front of each group
    const int32_t* values = static_cast<const int32_t*>(s.data) + slice.row_begin;
    constexpr uint32_t GROUPS = SLICE_ROWS / 1024;
    u32x4_t group[GROUPS];
    int32_t front[GROUPS];
#pragma unroll
    for (uint32_t i = 0; i < GROUPS; ++i) {
      const uint32_t first = i * 1024 + tid * 4;
      group[i] = *reinterpret_cast<const u32x4_t*>(values + (first < slice.row_count ? first : 0));
      front[i] = values[first > 0 && first < slice.row_count ? first - 1 : 0];
    }
#pragma unroll
    for (uint32_t i = 0; i < GROUPS; ++i) {
      const uint32_t first = i * 1024 + tid * 4;
      const int32_t k[5] = {front[i], static_cast<int32_t>(group[i].x), static_cast<int32_t>(group[i].y), static_cast<int32_t>(group[i].z), static_cast<int32_t>(group[i].w)};
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        if (first + e >= slice.row_count) continue;
        const uint64_t previous = first + e == 0 ? word_before : (static_cast<uint64_t>(static_cast<int64_t>(k[e])) - key_min) >> 5;
        place(k[e + 1], previous, first + e);
      }
    }
  } else {
  constexpr uint32_t BATCH = 8;
#pragma unroll 1
  for (uint32_t block = 0; block < SLICE_ROWS / 256 / BATCH; ++block) {
    if (block * BATCH * 256 >= slice.row_count) break;
    int64_t key[BATCH], before[BATCH];
#pragma unroll
    for (uint32_t i = 0; i < BATCH; ++i) {
      const uint32_t r = (block * BATCH + i) * 256 + tid;
      const uint32_t row = slice.row_begin + (r < slice.row_count ? r : 0);
      key[i] = dense_key(s, row);
      before[i] = dense_key(s, row > slice.row_begin ? row - 1 : row);
    }
#pragma unroll
    for (uint32_t i = 0; i < BATCH; ++i) {
      const uint32_t r = (block * BATCH + i) * 256 + tid;
      if (r >= slice.row_count) continue;
      place(key[i], r == 0 ? word_before : (static_cast<uint64_t>(before[i]) - key_min) >> 5, r);
    }
  }
  }
  if (!staged) return;
  __syncthreads();
  for (uint32_t i = tid; i < span; i += 256) {
    const uint32_t bits = s_bits[i], base = s_base[i];
    if (i == 0 || i + 1 == span) {   // may be shared with a neighbouring slice: add the bits; the base comes from the slice with the word's first key
      if (bits) atomicOr(reinterpret_cast<uint32_t*>(entries + first_word + i), bits);
      if (base) reinterpret_cast<uint32_t*>(entries + first_word + i)[1] = base - 1u;
    } else {
      entries[first_word + i] = u32x2_entry_t{bits, base ? base - 1u : 0u};
    }
  }
}

// Two regions zeroed by one launch with 16-byte stores (sizes in 16-byte vectors; pool blocks: aligned, and as long as their rounded
// sizes): the rank table and the Bloom filter of a hinted build -- two hipMemsetAsync calls were two launches and ~20 us of gaps.
__global__ __launch_bounds__(256) void zero_vectors(u32x4_t* first, size_t first_vectors, u32x4_t* second, size_t second_vectors) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < first_vectors + second_vectors; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    if (i < first_vectors) first[i] = u32x4_t{0, 0, 0, 0}; else second[i - first_vectors] = u32x4_t{0, 0, 0, 0};
  }
}

// Two small regions zeroed by one launch (a hipMemsetAsync each is ~5 us of stream time).
__global__ __launch_bounds__(256) void zero_two(uint32_t* first, size_t first_words, uint32_t* second, size_t second_words) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < first_words + second_words; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    if (i < first_words) first[i] = 0; else second[i - first_words] = 0;
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

// keys sorted ascending (unsigned bit order)?  Also: are two neighbours equal (a sorted column with duplicates), the OR of
// all keys (significant bytes for the radix sort) and the smallest / largest key in the order of the 64-bit key bits
// (the rank table's origin and extent): one round of atomics per 1024-thread workgroup, at most 1024 workgroups
// (atomics on one word retire at ~88 per microsecond).
// The directory keeps the keys in the unsigned order of their 64-bit bits; the rank table wants the SIGNED order (a key
// column with negative and positive values is dense there, not 2^63 apart): min / max are kept biased by the sign bit.
// flags (u32 words): [0] unsorted  [1] equal neighbours  [2] any NULL materialised  [3] unsorted in signed order
//                    [4..5] OR  [6..7] min ^ sign  [8..9] max ^ sign
template <typename K>
__global__ __launch_bounds__(1024) void check_sorted(const K* keys, uint64_t n, uint32_t* flags) {
  __shared__ uint64_t s_bits[16], s_min[16], s_max[16];
  __shared__ uint32_t s_unsorted, s_equal, s_unsorted_signed;
  if (threadIdx.x == 0) { s_unsorted = 0; s_equal = 0; s_unsorted_signed = 0; }
  __syncthreads();
  uint64_t bits = 0, lowest = ~0ull, highest = 0;
  bool unsorted_here = false, equal_here = false, unsorted_signed_here = false;
  constexpr uint64_t SIGN = 1ull << 63;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const K key = keys[i];
    bits |= key;
    const uint64_t wide = key_bits_of(key) ^ SIGN;
    lowest = wide < lowest ? wide : lowest;
    highest = wide > highest ? wide : highest;
    if (i + 1 < n) {
      const K next = keys[i + 1];
      if (key > next) unsorted_here = true;
      if (key == next) equal_here = true;
      if (wide > (key_bits_of(next) ^ SIGN)) unsorted_signed_here = true;
    }
  }
  if (unsorted_here) s_unsorted = 1;
  if (equal_here) s_equal = 1;
  if (unsorted_signed_here) s_unsorted_signed = 1;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    bits |= __shfl_xor(bits, d, 64);
    const uint64_t other_low = __shfl_xor(lowest, d, 64), other_high = __shfl_xor(highest, d, 64);
    lowest = other_low < lowest ? other_low : lowest;
    highest = other_high > highest ? other_high : highest;
  }
  if ((threadIdx.x & 63) == 0) { s_bits[threadIdx.x >> 6] = bits; s_min[threadIdx.x >> 6] = lowest; s_max[threadIdx.x >> 6] = highest; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t all = 0, low = ~0ull, high = 0;
    for (uint32_t w = 0; w < 16; ++w) { all |= s_bits[w]; low = s_min[w] < low ? s_min[w] : low; high = s_max[w] > high ? s_max[w] : high; }
    if (all) atomicOr(reinterpret_cast<unsigned long long*>(flags + 4), static_cast<unsigned long long>(all));
    atomicMin(reinterpret_cast<unsigned long long*>(flags + 6), static_cast<unsigned long long>(low));
    atomicMax(reinterpret_cast<unsigned long long*>(flags + 8), static_cast<unsigned long long>(high));
    if (s_unsorted) flags[0] = 1;
    if (s_equal) flags[1] = 1;
    if (s_unsorted_signed) flags[3] = 1;
  }
}

// ---- stable LSD radix sort of (key, RowID) pairs, 8 bits per pass ---------------------------------------------------
// histogram: [256][tiles] (digit-major so that one exclusive scan over the flat array yields scatter bases)
constexpr uint32_t SORT_TILE = 2048;
template <typename K>
__global__ __launch_bounds__(256) void sort_histogram(const K* keys, uint64_t n, uint32_t shift, uint32_t* hist, uint32_t n_tiles) {
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

// Inclusive prefix sum over the lanes of a wave (DPP; see scan.hip).
__device__ __forceinline__ uint32_t join_wave_inclusive_scan(uint32_t v) {
  asm volatile(
      "s_nop 4\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(v));
  return v;
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

// The same with all eight digit bits, unrolled (bits a digit never has: every lane agrees, the step changes nothing).
__device__ __forceinline__ uint64_t match_any8(uint32_t digit, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (uint32_t b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;   // (of a lane that is not valid: meaningless)
}

template <typename K, typename R>
__global__ __launch_bounds__(256) void sort_scatter(const K* keys_in, const R* rows_in, K* keys_out, R* rows_out,
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

// The same pass over 8192-element tiles with the tile's elements staged in LDS digit by digit: a digit's elements of a tile leave as one
// run per array (32 elements on average: whole 128-byte lines) instead of one scattered 4-byte store each -- sort_scatter's 2048-element
// tiles wrote 15 M keys and 15 M RowIDs per pass at 0.75 TB/s.  The rank inside (wave, digit) is one returning LDS atomic per element
// (lane-ordered: lds_atomic_order_probe; the host keeps sort_scatter where that does not hold).  uint32 keys and packed RowIDs.
constexpr uint32_t SORT_BIG_TILE = 8192, SORT_BIG_THREADS = 512, SORT_BIG_WAVES = SORT_BIG_THREADS / 64, SORT_BIG_ROUNDS = SORT_BIG_TILE / SORT_BIG_THREADS;
__global__ __launch_bounds__(SORT_BIG_THREADS) void sort_histogram_big(const uint32_t* keys, uint64_t n, uint32_t shift, uint32_t* hist, uint32_t n_tiles) {
  __shared__ uint32_t s_hist[4][256];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 4 * 256; i += SORT_BIG_THREADS) (&s_hist[0][0])[i] = 0;
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * SORT_BIG_TILE;
#pragma unroll
  for (uint32_t k = 0; k < SORT_BIG_ROUNDS; ++k) {
    const uint64_t i = base + k * SORT_BIG_THREADS + tid;
    if (i < n) atomicAdd(&s_hist[tid & 3][(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  if (tid < 256) hist[static_cast<size_t>(tid) * n_tiles + blockIdx.x] = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
}

__global__ __launch_bounds__(SORT_BIG_THREADS) void sort_scatter_staged(const uint32_t* keys_in, const uint32_t* rows_in, uint32_t* keys_out, uint32_t* rows_out, uint64_t n, uint32_t shift,
                                                                         const uint64_t* bases, uint32_t n_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sort_smem[];
  u32x2_t* s_stage = reinterpret_cast<u32x2_t*>(sort_smem);                 // [SORT_BIG_TILE]
  uint32_t* s_wave = sort_smem + 2 * SORT_BIG_TILE;                           // [SORT_BIG_WAVES][256]
  uint32_t* s_first = s_wave + SORT_BIG_WAVES * 256;                          // [256]
  uint64_t* s_base = reinterpret_cast<uint64_t*>(s_first + 256);              // [256]
  uint32_t* s_totals = reinterpret_cast<uint32_t*>(s_base + 256);             // [8]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t tile_base = static_cast<uint64_t>(blockIdx.x) * SORT_BIG_TILE;
  const uint32_t count = n - tile_base < SORT_BIG_TILE ? static_cast<uint32_t>(n - tile_base) : SORT_BIG_TILE;
  for (uint32_t i = tid; i < SORT_BIG_WAVES * 256; i += SORT_BIG_THREADS) s_wave[i] = 0;
  // wave w owns elements [w * 1024, (w + 1) * 1024) of the tile, element w * 1024 + k * 64 + lane in round k: element order = (wave, round, lane)
  uint32_t key[SORT_BIG_ROUNDS], row[SORT_BIG_ROUNDS], before[SORT_BIG_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < SORT_BIG_ROUNDS; ++k) {
    const uint32_t e = wave * (SORT_BIG_TILE / SORT_BIG_WAVES) + k * 64 + lane;
    key[k] = e < count ? keys_in[tile_base + e] : 0u;
    row[k] = e < count ? rows_in[tile_base + e] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < SORT_BIG_ROUNDS; ++k) {
    const uint32_t e = wave * (SORT_BIG_TILE / SORT_BIG_WAVES) + k * 64 + lane;
    before[k] = 0;
    if (e < count) before[k] = atomicAdd(&s_wave[wave * 256 + ((key[k] >> shift) & 0xFF)], 1u);
  }
  __syncthreads();
  uint32_t total = 0;
  if (tid < 256) {
#pragma unroll
    for (uint32_t w = 0; w < SORT_BIG_WAVES; ++w) { const uint32_t c = s_wave[w * 256 + tid]; s_wave[w * 256 + tid] = total; total += c; }
  }
  const uint32_t inclusive = join_wave_inclusive_scan(total);
  if (lane == 63) s_totals[wave] = inclusive;
  __syncthreads();
  uint32_t earlier_waves = 0;
  for (uint32_t w = 0; w < wave; ++w) earlier_waves += s_totals[w];
  if (tid < 256) {
    const uint32_t first = earlier_waves + inclusive - total;
    s_first[tid] = first;
    s_base[tid] = bases[static_cast<size_t>(tid) * n_tiles + blockIdx.x] - first;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < SORT_BIG_ROUNDS; ++k) {
    const uint32_t e = wave * (SORT_BIG_TILE / SORT_BIG_WAVES) + k * 64 + lane;
    if (e >= count) continue;
    const uint32_t digit = (key[k] >> shift) & 0xFF;
    s_stage[s_first[digit] + s_wave[wave * 256 + digit] + before[k]] = u32x2_t{key[k], row[k]};
  }
  __syncthreads();
  for (uint32_t i = tid; i < count; i += SORT_BIG_THREADS) {
    const u32x2_t element = s_stage[i];
    const uint64_t at = s_base[(element.x >> shift) & 0xFF] + i;
    keys_out[at] = element.x;
    rows_out[at] = element.y;
  }
}
__host__ __device__ constexpr size_t sort_staged_lds_bytes() { return 4 * (2 * size_t{SORT_BIG_TILE} + SORT_BIG_WAVES * 256 + 256 + 2 * 256 + 8); }

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

// `restart`: index of the block at which the running sum starts again from zero (two arrays scanned by one set of
// launches, the second one laid out from a block boundary), or 0xFFFFFFFF.  *total_out = the (last) segment's total.
__global__ __launch_bounds__(1024) void scan_block_offsets(uint64_t* block_sums, uint32_t n_blocks, uint32_t restart, uint64_t* total_out) {
  __shared__ uint64_t s_partial[1024];
  __shared__ uint64_t s_first_segment;
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n_blocks + 1023) / 1024;
  const uint32_t begin = tid * per < n_blocks ? tid * per : n_blocks, end = begin + per < n_blocks ? begin + per : n_blocks;
  uint64_t sum = 0;
  for (uint32_t i = begin; i < end; ++i) sum += block_sums[i];
  // exclusive prefix of the 1024 partial sums: inside every wave, then over the 16 wave totals
  const uint32_t lane = tid & 63, wave = tid >> 6;
  uint64_t inclusive = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(inclusive, d, 64);
    if (lane >= static_cast<uint32_t>(d)) inclusive += t;
  }
  if (lane == 63) s_partial[wave] = inclusive;
  __syncthreads();
  if (wave == 0) {
    const uint64_t total = lane < 16 ? s_partial[lane] : 0;
    uint64_t scanned = total;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint64_t t = __shfl_up(scanned, d, 64);
      if (lane >= static_cast<uint32_t>(d)) scanned += t;
    }
    if (lane < 16) s_partial[16 + lane] = scanned - total;
    if (lane == 15) *total_out = scanned;   // (with a restart: corrected below)
  }
  __syncthreads();
  uint64_t run = s_partial[16 + wave] + inclusive - sum;
  for (uint32_t i = begin; i < end; ++i) { const uint64_t v = block_sums[i]; block_sums[i] = run; run += v; }
  if (restart >= n_blocks) return;
  __syncthreads();
  if (tid == 0) s_first_segment = block_sums[restart];   // everything in front of the second segment
  __syncthreads();
  for (uint32_t i = begin > restart ? begin : restart; i < end; ++i) block_sums[i] -= s_first_segment;
  if (tid == 15) *total_out -= s_first_segment;   // (the lane that wrote it)
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

// ---- bucket directory over the sorted build keys ------------------------------------------------------------------------
struct Directory {
  const uint64_t* keys;      // sorted (unsigned order of the sign-extended bits); nullptr when the keys are 32-bit
  const uint32_t* keys32;    // int32 build columns instead: the keys' low halves (same order), padded by four entries
  const hy_row_id* row_ids;  // same order; nullptr when the RowIDs are packed
  const uint32_t* ids32;     // ... chunk_id << 16 | chunk_offset instead (every build RowID fits)
  const uint32_t* dir;       // [n_buckets + 1] first position of every bucket
  uint64_t n;
  uint64_t key_min;
  uint64_t key_max;
  uint32_t shift;
  uint32_t n_buckets;
};

__device__ __forceinline__ uint64_t directory_key(const Directory& d, uint32_t i) {
  return d.keys32 ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(d.keys32[i]))) : d.keys[i];
}
__device__ __forceinline__ hy_row_id directory_row_id(const Directory& d, uint32_t i) {
  if (d.ids32) { const uint32_t id = d.ids32[i]; return hy_row_id{id >> 16, id & 0xFFFFu}; }
  return d.row_ids[i];
}
template <typename K> __device__ __forceinline__ uint64_t key_bits(K key) {   // the 64-bit (sign-extended) key of a stored build key
  if constexpr (sizeof(K) == 4) return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(key)));
  else return key;
}

template <typename K>
__global__ void directory_fill(const K* keys, uint64_t n, uint64_t key_min, uint32_t shift, uint32_t n_buckets, uint32_t* dir) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t bucket = (key_bits(keys[i]) - key_min) >> shift;
  const int64_t previous = i == 0 ? -1 : static_cast<int64_t>((key_bits(keys[i - 1]) - key_min) >> shift);
  for (int64_t b = previous + 1; b <= static_cast<int64_t>(bucket); ++b) dir[b] = static_cast<uint32_t>(i);
  if (i + 1 == n) {
    for (uint64_t b = bucket + 1; b <= n_buckets; ++b) dir[b] = static_cast<uint32_t>(n);
  }
}

// (start, count) of `key` in the sorted build keys.  Buckets of up to four keys (the directory is sized for ~one key
// per bucket) are probed with four independent loads instead of a dependent binary search chain.
__device__ __forceinline__ void directory_lookup(const Directory& d, uint64_t key, uint32_t* start, uint32_t* count) {
  *start = 0;
  *count = 0;
  if (d.n == 0 || key < d.key_min || key > d.key_max) return;
  const uint64_t bucket = (key - d.key_min) >> d.shift;
  uint32_t lo = d.dir[bucket], hi = d.dir[bucket + 1];
  if (hi - lo <= 4) {
    const uint32_t last = static_cast<uint32_t>(d.n - 1);
    const uint64_t k0 = directory_key(d, lo < last ? lo : last), k1 = directory_key(d, lo + 1 < last ? lo + 1 : last), k2 = directory_key(d, lo + 2 < last ? lo + 2 : last),
                   k3 = directory_key(d, lo + 3 < last ? lo + 3 : last);
    const uint32_t size = hi - lo;
    const uint32_t equal = (size > 0 && k0 == key ? 1u : 0u) | (size > 1 && k1 == key ? 2u : 0u) | (size > 2 && k2 == key ? 4u : 0u) | (size > 3 && k3 == key ? 8u : 0u);
    if (equal) {   // equal keys are adjacent
      *start = lo + (__ffs(equal) - 1);
      *count = __popc(equal);
    }
    return;
  }
  const uint32_t bucket_end = hi;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (directory_key(d, mid) < key) lo = mid + 1; else hi = mid;
  }
  if (lo == bucket_end || directory_key(d, lo) != key) return;
  *start = lo;
  uint32_t end = lo + 1;
  if (end < bucket_end && directory_key(d, end) == key) {   // duplicates: upper bound inside the bucket
    uint32_t l = end, h = bucket_end;
    while (l < h) {
      const uint32_t mid = l + (h - l) / 2;
      if (directory_key(d, mid) <= key) l = mid + 1; else h = mid;
    }
    end = l;
  }
  *count = end - lo;
}

// ---- rank table over the build keys ------------------------------------------------------------------------------------
// Integer build keys that are unique and not too sparse (primary keys: TPC-H o_orderkey uses 8 of every 32 integers, SSB's
// dimension keys are dense) get a succinct rank dictionary instead of the bucket directory: one 8-byte entry per 32
// consecutive key values, {bit per value present, number of build keys below the entry's first value}.  A probe is ONE
// dependent 8-byte load: the key is a build key iff its bit is set, and its rank among the build keys -- its position in
// the key-ordered build side -- is base + popcount(bits below).  No key is ever compared, so the sorted key array is not
// read by the probe at all; the table (range / 4 bytes: 15 MB for SF10 orders) stays in the L2s / Infinity Cache where
// the directory and its key array (120 MB) cannot.  The same open-addressing idea as the directory with a hash function
// that is the key itself.
struct RankTable {
  const u32x2_t* entries;    // [(range >> 5) + 1]
  uint64_t key_min;
  uint64_t range;            // key_max - key_min
  uint32_t identity_rows;    // != 0: the build row of rank r is RowID{r / identity_rows, r % identity_rows} (a dense, sorted build
  uint32_t reserved;         //       column whose chunks all hold identity_rows rows): no RowID array at all
  double identity_inverse;   // 1.0 / identity_rows
};

// Sorted, duplicate-free keys: the first key of every 32-value word collects the word's bits (its followers are the next
// <= 31 keys) and stores the entry -- no atomics, no scan; words without a key keep their zeros.
template <typename K>
__global__ __launch_bounds__(256) void rank_table_fill_sorted(const K* keys, uint64_t n, uint64_t key_min, u32x2_t* entries) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t rel = key_bits_of(keys[i]) - key_min;
  const uint64_t word = rel >> 5;
  if (i != 0 && ((key_bits_of(keys[i - 1]) - key_min) >> 5) == word) return;
  uint32_t bits = 1u << (rel & 31);
  for (uint64_t j = i + 1; j < n && j < i + 32; ++j) {
    const uint64_t other = key_bits_of(keys[j]) - key_min;
    if ((other >> 5) != word) break;
    bits |= 1u << (other & 31);
  }
  entries[word] = u32x2_t{bits, static_cast<uint32_t>(i)};
}

// keys[i] <- keys[i] + delta (uint32 arithmetic): a build side's keys as distances from the smallest one and back
__global__ __launch_bounds__(256) void shift_keys(uint32_t* keys, uint64_t n, uint32_t delta) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) keys[i] += delta;
}

// Unsorted keys: every key sets its bit (a bit that is already set: the keys are not unique, the join falls back to the
// sorted directory), a scan over the words' population counts gives the bases, and the RowIDs are scattered to their
// keys' ranks -- which is all the "sort" a unique build side needs.
template <typename K>
__global__ __launch_bounds__(256) void rank_table_mark(const K* keys, uint64_t n, uint64_t key_min, u32x2_t* entries, uint32_t* duplicate) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t rel = key_bits_of(keys[i]) - key_min;
  const uint32_t bit = 1u << (rel & 31);
  const uint32_t old = atomicOr(reinterpret_cast<uint32_t*>(entries + (rel >> 5)), bit);
  if (old & bit) *duplicate = 1;
}

constexpr uint32_t RANK_BLOCK = 4096;   // table words per workgroup of the base scan
__global__ __launch_bounds__(256) void rank_table_block_sums(const u32x2_t* entries, uint64_t words, uint64_t* block_sums) {
  __shared__ uint32_t s_wave[4];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * RANK_BLOCK;
  uint32_t sum = 0;
  for (uint32_t k = 0; k < RANK_BLOCK / 256; ++k) {
    const uint64_t w = base + k * 256 + threadIdx.x;
    if (w < words) sum += __popc(entries[w].x);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = uint64_t{s_wave[0]} + s_wave[1] + s_wave[2] + s_wave[3];
}

__global__ __launch_bounds__(256) void rank_table_bases(u32x2_t* entries, uint64_t words, const uint64_t* block_offsets) {
  __shared__ uint32_t s_wave[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t first = static_cast<uint64_t>(blockIdx.x) * RANK_BLOCK + static_cast<uint64_t>(tid) * (RANK_BLOCK / 256);
  uint32_t counts[RANK_BLOCK / 256];
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < RANK_BLOCK / 256; ++k) {
    counts[k] = first + k < words ? __popc(entries[first + k].x) : 0;
    sum += counts[k];
  }
  const uint32_t inclusive = join_wave_inclusive_scan(sum);
  if (lane == 63) s_wave[wave] = inclusive;
  __syncthreads();
  uint32_t run = static_cast<uint32_t>(block_offsets[blockIdx.x]) + inclusive - sum;
  for (uint32_t w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
  for (uint32_t k = 0; k < RANK_BLOCK / 256; ++k) {
    if (first + k < words) reinterpret_cast<uint32_t*>(entries + first + k)[1] = run;
    run += counts[k];
  }
}

template <typename K, typename R>
__global__ __launch_bounds__(256) void rank_table_scatter_rows(const K* keys, const R* rows_in, R* rows_out, uint64_t n, uint64_t key_min, const u32x2_t* entries) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t rel = key_bits_of(keys[i]) - key_min;
  const u32x2_t entry = entries[rel >> 5];
  rows_out[entry.y + __popc(entry.x & ((1u << (rel & 31)) - 1))] = rows_in[i];
}

// ---- probe side ------------------------------------------------------------------------------------------------------
struct JoinPlan {   // written by plan_output between the probe passes: what pass 2's kernels need to know
  uint32_t fits, n_slices;   // the result buffers hold the pairs and PosLists | number of output PosLists
};

// A secondary join predicate as the probe sees it:  build_value <condition> probe_value  (the caller's
// left <condition> right, flipped when the build side is the right input: join_hash.cpp:158-165).
struct SecondaryPredicate {
  const DevSegment* build;
  const DevSegment* probe;
  uint32_t condition;
  uint32_t reserved;
};

struct ProbeArgs {
  const DevSegment* segments;     // probe column
  const Slice* slices;            // 8192-row slices; a tile is half a slice
  const SliceView* views;         // the same slices for the kernels that fetch ahead (rt_probe_count / rt_probe_emit)
  uint32_t n_tiles;
  uint32_t mode;                  // HY_JOIN_*
  uint32_t radix_bits;
  uint32_t keep_nulls;            // probe side keeps NULLs (Left/Right/Anti*)
  uint32_t build_rows_zero;       // build table has no rows (AntiNullAsTrue special case)
  const uint8_t* build_bloom;     // filter applied to the probe side, or nullptr
  Directory dir;
  RankTable rank;                 // rank.entries != nullptr: the rt_* kernels look keys up here (dir then only carries the RowIDs by rank)
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
  uint32_t* row_partner;          // [n_tiles][JOIN_TILE] pass 1 -> pass 2: build position of the row's partner, or nullptr
  uint32_t* row_meta;             // [n_tiles][JOIN_WAVES][JOIN_ROUNDS / 2][64] pass 1 -> pass 2: partition and flags (ROW_*) of a lane's rows, two rounds per word
  uint32_t* tile_uncached;        // [n_tiles] set by pass 1 when a tile has a row with several partners (pass 2 evaluates it again)
  uint32_t* n_uncached;           // number of such tiles
  uint32_t* uncached_tiles;       // [n_tiles] ... and which (in no particular order)
  uint32_t* xcd_tickets;          // [8] probe_emit_cached: next tile of every XCD's share
  const JoinPlan* plan;           // pass 2: does the result fit its buffers, how many output PosLists
  uint32_t n_secondary;           // secondary predicates: every partner the key lookup finds is tested against them
  SecondaryPredicate secondary[HY_MAX_SECONDARY_PREDICATES];
  uint32_t* error;                // set when a probe row matches >= 2^22 build rows (the staging record cannot hold it)
  uint64_t* trace;                // debug (HY_JOIN_TRACE): 6 wall-clock stamps per probe_emit tile, else nullptr
  uint32_t pack_build_ids;        // dir.ids32 exists: probe_emit_cached stages the partner's packed RowID, not its position
  uint32_t hashed_type;           // 0: integer keys | HY_TYPE_FLOAT | HY_TYPE_DOUBLE (join_hash); only the <true> instantiations look at it
  uint32_t lane_ordered_atomics;  // rt_probe_emit: a returning LDS atomic hands the lanes of one instruction their values in lane order (probed once per process)
};

// Output pairs of one probe row per join mode (probe / probe_semi_anti, join_hash_steps.hpp:575-922).
__device__ __forceinline__ uint32_t pairs_of(const ProbeArgs& a, bool is_null, uint32_t count, bool* null_partner) {
  // Selects on the (uniform) mode instead of a switch: this runs per probe row, eight times unrolled -- a switch is half
  // a dozen scalar branches each time.
  const uint32_t mode = a.mode;
  const bool outer = mode == HY_JOIN_LEFT || mode == HY_JOIN_RIGHT;
  const bool none = is_null || count == 0;
  *null_partner = outer && none;
  const uint32_t anti_null_as_true = is_null ? (a.build_rows_zero ? 1u : 0u) : (count == 0 ? 1u : 0u);
  return mode == HY_JOIN_INNER ? count
         : outer ? (none ? 1u : count)
         : mode == HY_JOIN_SEMI ? (count > 0 ? 1u : 0u)
         : mode == HY_JOIN_ANTI_NULL_AS_FALSE ? (none ? 1u : 0u)
         : anti_null_as_true;
}

// ---- secondary predicates (MultiPredicateJoinEvaluator, multi_predicate_join_evaluator.hpp:30-61) -------------------------
// x <condition> y in the common C++ type of the two column types: the reference's comparator functors are generic
// lambdas, the usual arithmetic conversions apply (int64 against float compares as float).
__device__ __forceinline__ bool compare_typed(uint32_t condition, const Value& x, uint32_t xt, const Value& y, uint32_t yt) {
  const bool x_float = xt == HY_TYPE_FLOAT || xt == HY_TYPE_DOUBLE, y_float = yt == HY_TYPE_FLOAT || yt == HY_TYPE_DOUBLE;
  bool less, equal;
  if (xt == HY_TYPE_DOUBLE || yt == HY_TYPE_DOUBLE) {
    const double p = x_float ? x.f : static_cast<double>(x.i), q = y_float ? y.f : static_cast<double>(y.i);
    less = p < q; equal = p == q;
  } else if (xt == HY_TYPE_FLOAT || yt == HY_TYPE_FLOAT) {
    const float p = x_float ? static_cast<float>(x.f) : static_cast<float>(x.i), q = y_float ? static_cast<float>(y.f) : static_cast<float>(y.i);
    less = p < q; equal = p == q;
  } else {
    less = x.i < y.i; equal = x.i == y.i;
  }
  switch (condition) {
    case HY_PRED_EQUALS: return equal;
    case HY_PRED_NOT_EQUALS: return !equal;
    case HY_PRED_LESS_THAN: return less;
    case HY_PRED_LESS_THAN_EQUALS: return less || equal;
    case HY_PRED_GREATER_THAN: return !less && !equal;
    default: return !less;   // HY_PRED_GREATER_THAN_EQUALS
  }
}

// Does the pair (build row, probe row) satisfy every secondary predicate?  A NULL on either side does not (:50-52).
__device__ __forceinline__ bool satisfies_secondary(const ProbeArgs& a, hy_row_id build_row, uint32_t probe_chunk, uint32_t probe_row) {
  for (uint32_t p = 0; p < a.n_secondary; ++p) {
    const SecondaryPredicate& predicate = a.secondary[p];
    const Value x = column_value(predicate.build, build_row.chunk_id, build_row.chunk_offset);
    const Value y = column_value(predicate.probe, probe_chunk, probe_row);
    if (x.is_null || y.is_null) return false;
    if (!compare_typed(predicate.condition, x, predicate.build[build_row.chunk_id].data_type, y, predicate.probe[probe_chunk].data_type)) return false;
  }
  return true;
}

// ---- batched evaluation: the JOIN_ROUNDS rows of a lane, phase by phase --------------------------------------------------
// A probe is a chain of dependent loads (key -> Bloom byte -> directory entry -> build keys).  Evaluating the lane's rows
// one after the other would pay that latency JOIN_ROUNDS times; here every phase issues the loads of all rows first.
// All loads are unconditional (indices of rows that do not exist are clamped to an existing row and the results masked
// afterwards): straight-line code, every phase's loads back to back, no exec-mask juggling around each load.
template <typename T>
__device__ __forceinline__ void load_rows(const void* data, const uint32_t (&index)[JOIN_ROUNDS], T (&out)[JOIN_ROUNDS]) {
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) out[k] = static_cast<const T*>(data)[index[k]];
}

__device__ __forceinline__ void load_compressed_rows(const void* data, uint32_t width, const uint32_t (&index)[JOIN_ROUNDS], uint32_t (&out)[JOIN_ROUNDS]) {
  if (width == 1) {
    uint8_t v[JOIN_ROUNDS];
    load_rows<uint8_t>(data, index, v);
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) out[k] = v[k];
  } else if (width == 2) {
    uint16_t v[JOIN_ROUNDS];
    load_rows<uint16_t>(data, index, v);
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) out[k] = v[k];
  } else {
    load_rows<uint32_t>(data, index, out);
  }
}

// Phase 1 of a probe: the keys of the lane's JOIN_ROUNDS rows (NULL rows: key 0, a kept NULL lands in partition 0).
template <bool GENERAL>   // the general instantiation also reads float / double keys, and integer keys cast to them
__device__ __forceinline__ void decode_keys(const ProbeArgs& a, uint32_t chunk, uint32_t row_begin, uint32_t row_count, uint32_t wave, uint32_t lane,
                                            uint32_t (&row)[JOIN_ROUNDS], bool (&in)[JOIN_ROUNDS], bool (&is_null)[JOIN_ROUNDS], int64_t (&key)[JOIN_ROUNDS]) {
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    const uint32_t r = wave * JOIN_WAVE_ROWS + k * 64 + lane;
    in[k] = r < row_count;
    row[k] = row_begin + (in[k] ? r : 0);   // row_count > 0: the tile's first row exists
    is_null[k] = false;
    key[k] = 0;
  }
  if constexpr (GENERAL) {
    if (a.hashed_type) {
#pragma unroll 1
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
        if (in[k]) is_null[k] = hashed_key(a.segments, chunk, row[k], a.hashed_type, &key[k]);
      }
      return;
    }
  }
  const DevSegment s = a.segments[chunk];
  if (s.encoding == HY_ENC_REFERENCE) {
#pragma unroll 1
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      if (in[k]) is_null[k] = column_key(a.segments, chunk, row[k], &key[k]);
    }
  } else if (s.encoding == HY_ENC_DICTIONARY) {
    uint32_t vid[JOIN_ROUNDS];
    load_compressed_rows(s.data, s.width, row, vid);
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      is_null[k] = vid[k] >= s.aux_size;
      if (is_null[k]) vid[k] = 0;
    }
    if (s.aux_size != 0) {
      if (s.data_type == HY_TYPE_INT) {
        int32_t v[JOIN_ROUNDS];
        load_rows<int32_t>(s.aux, vid, v);
#pragma unroll
        for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) key[k] = v[k];
      } else {
        load_rows<int64_t>(s.aux, vid, key);
      }
    }
  } else {
    if (s.nulls) {
      uint32_t word[JOIN_ROUNDS];
      uint64_t bits[JOIN_ROUNDS];
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) word[k] = row[k] >> 6;
      load_rows<uint64_t>(s.nulls, word, bits);
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) is_null[k] = (bits[k] >> (row[k] & 63)) & 1;
    }
    if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
      uint32_t raw[JOIN_ROUNDS];
      // A wave's rows (JOIN_WAVE_ROWS consecutive ones, starting at a multiple of JOIN_WAVE_ROWS inside the chunk: tiles are halves
      // of 8192-row slices) lie in ONE 2048-row block: its minimum is a scalar load, not eight vector loads per lane.
      static_assert(HY_FOR_BLOCK_SIZE % JOIN_WAVE_ROWS == 0 && JOIN_TILE % HY_FOR_BLOCK_SIZE == 0, "a wave's rows must not straddle FrameOfReference blocks");
      const uint32_t wave_row = __builtin_amdgcn_readfirstlane(row_begin + (wave * JOIN_WAVE_ROWS < row_count ? wave * JOIN_WAVE_ROWS : 0u));
      const uint32_t bias = static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[wave_row / HY_FOR_BLOCK_SIZE]);
      if (row_count == JOIN_TILE) {   // a full tile: row k of the lane is 64 elements behind row k - 1 -- one address, immediate offsets
        const uint32_t first = row_begin + wave * JOIN_WAVE_ROWS + lane;
        if (s.width == 2) {
          const uint16_t* base = static_cast<const uint16_t*>(s.data) + first;
#pragma unroll
          for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) raw[k] = base[k * 64];
        } else if (s.width == 1) {
          const uint8_t* base = static_cast<const uint8_t*>(s.data) + first;
#pragma unroll
          for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) raw[k] = base[k * 64];
        } else {
          const uint32_t* base = static_cast<const uint32_t*>(s.data) + first;
#pragma unroll
          for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) raw[k] = base[k * 64];
        }
      } else {
        load_compressed_rows(s.data, s.width, row, raw);
      }
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) key[k] = static_cast<int32_t>(raw[k] + bias);
    } else if (s.data_type == HY_TYPE_INT) {
      int32_t v[JOIN_ROUNDS];
      load_rows<int32_t>(s.data, row, v);
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) key[k] = v[k];
    } else {
      load_rows<int64_t>(s.data, row, key);
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    if (is_null[k]) key[k] = 0;
  }
}

// What pass 1 leaves behind for pass 2, per probe row: a 16-bit word
//   [7:0] radix partition  [8] materialised  [9] one output pair  [10] its build side is NULL_ROW_ID
// (stored lane-major, the words of rounds 2j and 2j+1 of a lane in one 32-bit word) and the partner's position in the
// sorted build side.  A tile with a row that has several partners is flagged and evaluated again by probe_emit_generic.
constexpr uint32_t ROW_PARTITION = 0xFF, ROW_MATERIALISED = 0x100, ROW_EMIT = 0x200, ROW_NULL_PARTNER = 0x400;

// meta[k] = emit << 10 | null_partner << 9 | partition (INVALID_PARTITION: not materialised); start[k] = first build position
template <bool SECONDARY>   // the general instantiation (<true>: secondary predicates, float / double keys) and the plain one, which pays nothing for them
__device__ __forceinline__ void evaluate_rows(const ProbeArgs& a, uint32_t chunk, uint32_t row_begin, uint32_t row_count, uint32_t wave, uint32_t lane,
                                              uint32_t (&meta)[JOIN_ROUNDS], uint32_t (&start)[JOIN_ROUNDS], uint32_t (&partners)[JOIN_ROUNDS]) {
  uint32_t row[JOIN_ROUNDS];
  bool in[JOIN_ROUNDS], is_null[JOIN_ROUNDS];
  int64_t key[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    meta[k] = INVALID_PARTITION;
    start[k] = 0;
    if constexpr (SECONDARY) partners[k] = 0;   // (only read with secondary predicates)
  }
  if (row_count == 0) return;
  decode_keys<SECONDARY>(a, chunk, row_begin, row_count, wave, lane, row, in, is_null, key);
  uint32_t hashed_type = 0;   // (constant 0 in the plain instantiation: std::hash is the identity and nothing below changes)
  if constexpr (SECONDARY) hashed_type = a.hashed_type;
  // ---- phase 2: which rows are materialised by the NULL policy (the Bloom filter follows the lookup, see phase 4)
  bool valid[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) valid[k] = in[k] && !(is_null[k] && !a.keep_nulls);
  // ---- phase 3: directory entries (dir[bucket], dir[bucket + 1] in one 8-byte load)
  const Directory& d = a.dir;
  bool look[JOIN_ROUNDS];
  uint32_t lo[JOIN_ROUNDS], hi[JOIN_ROUNDS], count[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    look[k] = false;
    lo[k] = hi[k] = count[k] = 0;
  }
  if (d.n != 0) {
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      const uint64_t hash = static_cast<uint64_t>(key[k]);
      look[k] = valid[k] && !is_null[k] && hash >= d.key_min && hash <= d.key_max && !key_is_nan(hash, hashed_type);
      const uint64_t bucket = look[k] ? (hash - d.key_min) >> d.shift : 0;
      const u32x2_t entry = *reinterpret_cast<const u32x2_t __attribute__((aligned(4)))*>(d.dir + bucket);
      lo[k] = entry.x;
      hi[k] = entry.y;
    }
    // ---- phase 4: the bucket's keys, four at a time (longer buckets fall back to a binary search)
    if (d.keys32) {   // int32 build keys: one 16-byte load per row (the array is padded by four entries)
      u32x4_t probe[JOIN_ROUNDS];
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) probe[k] = *reinterpret_cast<const u32x4_t __attribute__((aligned(4)))*>(d.keys32 + lo[k]);
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
        const uint32_t size = hi[k] - lo[k];
        const bool fits = key[k] == static_cast<int64_t>(static_cast<int32_t>(key[k]));   // outside int32: no partner
        const uint32_t want = static_cast<uint32_t>(key[k]);
        const uint32_t equal = (size > 0 && probe[k].x == want ? 1u : 0u) | (size > 1 && probe[k].y == want ? 2u : 0u) | (size > 2 && probe[k].z == want ? 4u : 0u) |
                               (size > 3 && probe[k].w == want ? 8u : 0u);
        if (look[k] && fits) {
          if (size <= 4) {
            if (equal) {   // equal keys are adjacent
              start[k] = lo[k] + (__ffs(equal) - 1);
              count[k] = __popc(equal);
            }
          } else {
            directory_lookup(d, static_cast<uint64_t>(key[k]), &start[k], &count[k]);
          }
        }
      }
    } else {
      const uint32_t last = static_cast<uint32_t>(d.n - 1);
#pragma unroll
      for (uint32_t half = 0; half < JOIN_ROUNDS; half += JOIN_ROUNDS / 2) {
        uint64_t probe[JOIN_ROUNDS / 2][4];
#pragma unroll
        for (uint32_t k = 0; k < JOIN_ROUNDS / 2; ++k) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) probe[k][j] = d.keys[lo[half + k] + j < last ? lo[half + k] + j : last];
        }
#pragma unroll
        for (uint32_t k = 0; k < JOIN_ROUNDS / 2; ++k) {
          const uint32_t i = half + k;
          const uint64_t hash = static_cast<uint64_t>(key[i]);
          const uint32_t size = hi[i] - lo[i];
          const uint32_t equal = (size > 0 && probe[k][0] == hash ? 1u : 0u) | (size > 1 && probe[k][1] == hash ? 2u : 0u) | (size > 2 && probe[k][2] == hash ? 4u : 0u) |
                                 (size > 3 && probe[k][3] == hash ? 8u : 0u);
          if (look[i]) {
            if (size <= 4) {
              if (equal) {
                start[i] = lo[i] + (__ffs(equal) - 1);
                count[i] = __popc(equal);
              }
            } else {
              directory_lookup(d, hash, &start[i], &count[i]);
            }
          }
        }
      }
    }
  }
  // The build side's Bloom filter decides which probe rows count as materialised (join_hash_steps.hpp:354-358).  Every
  // build key is in the filter, so only rows WITHOUT a partner need the test: usually none of a wave's rows.
  if (a.build_bloom && !a.keep_nulls) {
    bool any = false;
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) any = any || (valid[k] && count[k] == 0);
    if (__any(any)) {
      uint32_t index[JOIN_ROUNDS];
      uint8_t hit[JOIN_ROUNDS];
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) index[k] = static_cast<uint32_t>(join_hash(static_cast<uint64_t>(key[k]), hashed_type)) & (BLOOM_BITS - 1);
      load_rows<uint8_t>(a.build_bloom, index, hit);
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) valid[k] = valid[k] && !(count[k] == 0 && hit[k] == 0);
    }
  }
  // ---- secondary predicates: of the partners the key found, the ones that also satisfy them (join_hash_steps.hpp:727-747,
  //      869-876).  partners[] keeps the key's partner count: pass 2 walks them again and writes the ones that pass.
  uint32_t passing[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) passing[k] = count[k];
  if constexpr (SECONDARY) {
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) partners[k] = count[k];
#pragma unroll 1
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      uint32_t n = 0;
      if (valid[k] && !is_null[k]) {
        for (uint32_t t = 0; t < count[k]; ++t) n += satisfies_secondary(a, directory_row_id(d, start[k] + t), chunk, row[k]) ? 1u : 0u;
      }
      passing[k] = n;
    }
  }
  // ---- phase 5: pairs per mode
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    if (!valid[k]) continue;
    const uint32_t partition = a.radix_bits ? static_cast<uint32_t>(join_hash(static_cast<uint64_t>(key[k]), hashed_type) & ((1u << a.radix_bits) - 1)) : 0;
    bool null_partner = false;
    uint32_t emit = pairs_of(a, is_null[k], passing[k], &null_partner);
    if (emit >= (1u << 22)) { *a.error = 1; emit = (1u << 22) - 1; }
    meta[k] = (emit << 10) | (null_partner ? 0x200u : 0u) | partition;
  }
}

// Workgroup -> tile.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): XCD x takes the x-th eighth
// of the tiles, so neighbouring tiles -- whose (tile, partition) output runs are neighbours in memory and whose directory
// and build-key lines overlap -- meet in the same L2 at about the same time.  The grid is 8 * ceil(n_tiles / 8).
__host__ __device__ constexpr uint32_t probe_grid(uint32_t n_tiles) { return 8 * ((n_tiles + 7) / 8); }
__device__ __forceinline__ uint32_t block_tile(uint32_t n_tiles) { return (blockIdx.x & 7) * ((n_tiles + 7) / 8) + (blockIdx.x >> 3); }

__device__ __forceinline__ void tile_rows(const ProbeArgs& a, uint32_t tile, uint32_t* chunk, uint32_t* row_begin, uint32_t* row_count) {
  const Slice slice = a.slices[tile / (SLICE_ROWS / JOIN_TILE)];
  const uint32_t offset = (tile % (SLICE_ROWS / JOIN_TILE)) * JOIN_TILE;
  *chunk = slice.chunk;
  *row_begin = slice.row_begin + offset;
  *row_count = slice.row_count > offset ? (slice.row_count - offset < JOIN_TILE ? slice.row_count - offset : JOIN_TILE) : 0;
}

// Pass 1: per tile (4096 consecutive probe rows) and radix partition, the number of materialised probe elements and of
// output pairs.  Wave w owns rows [w*512, (w+1)*512) of the tile, 64 consecutive rows per round.
template <bool SECONDARY>
__global__ __launch_bounds__(JOIN_THREADS) void probe_count(ProbeArgs a) {
  __shared__ uint32_t s_elements[MAX_PARTITIONS + 1];   // + 1: "a row of this tile has several partners"
  __shared__ uint32_t s_pairs[MAX_PARTITIONS];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t tile = block_tile(a.n_tiles);
  if (tile >= a.n_tiles) return;
  if (tid < MAX_PARTITIONS) { s_elements[tid] = 0; s_pairs[tid] = 0; }
  if (tid == 0) s_elements[MAX_PARTITIONS] = 0;
  __syncthreads();
  uint32_t chunk, row_begin, row_count;
  tile_rows(a, tile, &chunk, &row_begin, &row_count);
  uint32_t meta[JOIN_ROUNDS], start[JOIN_ROUNDS], partners[JOIN_ROUNDS];
  evaluate_rows<SECONDARY>(a, chunk, row_begin, row_count, wave, lane, meta, start, partners);
  bool fits = !SECONDARY;   // (with secondary predicates pass 2 has to test the partners again: probe_emit_generic)
#pragma unroll
  for (uint32_t round = 0; round < JOIN_ROUNDS; ++round) {
    const uint32_t partition = meta[round] & 0x1FF;
    const uint32_t emit = meta[round] >> 10;
    if (partition != INVALID_PARTITION) {
      atomicAdd(&s_elements[partition], 1u);
      if (emit) atomicAdd(&s_pairs[partition], emit);
    }
    fits = fits && emit <= 1;
  }
  if (a.row_meta) {
    uint32_t word[JOIN_ROUNDS];
#pragma unroll
    for (uint32_t round = 0; round < JOIN_ROUNDS; ++round) {
      const uint32_t partition = meta[round] & 0x1FF;
      word[round] = partition == INVALID_PARTITION ? 0u
                                                   : partition | ROW_MATERIALISED | ((meta[round] >> 10) ? ROW_EMIT : 0u) | ((meta[round] & 0x200u) ? ROW_NULL_PARTNER : 0u);
      if (word[round] & ROW_EMIT) __builtin_nontemporal_store(start[round], a.row_partner + static_cast<size_t>(tile) * JOIN_TILE + wave * JOIN_WAVE_ROWS + round * 64 + lane);
    }
#pragma unroll
    for (uint32_t j = 0; j < JOIN_ROUNDS / 2; ++j)   // (rows past the tile's end: 0, pass 2 loads whole tiles)
      __builtin_nontemporal_store(word[2 * j] | word[2 * j + 1] << 16, a.row_meta + static_cast<size_t>(tile) * (JOIN_TILE / 2) + (wave * (JOIN_ROUNDS / 2) + j) * 64 + lane);
    if (!__all(fits) && lane == 0) s_elements[MAX_PARTITIONS] = 1;
  }
  __syncthreads();
  if (a.row_meta && tid == 0) {
    a.tile_uncached[tile] = s_elements[MAX_PARTITIONS];
    if (s_elements[MAX_PARTITIONS]) a.uncached_tiles[atomicAdd(a.n_uncached, 1u)] = tile;   // probe_emit_generic's work list
  }
  if (tid < partitions) {
    a.hist_elements[static_cast<size_t>(tid) * a.n_tiles + tile] = s_elements[tid];
    a.hist_pairs[static_cast<size_t>(tid) * a.n_tiles + tile] = s_pairs[tid];
  }
}

// ---- pass 2, common case: every row of the tile has at most one partner ------------------------------------------------
// Persistent workgroups (three per CU), each walking its share of an XCD's tiles with the NEXT tile's loads in flight:
// pass 1 left a 16-bit word and the partner per row, so a tile is two plain coalesced loads per row -- no segment
// descriptors, no key decoding, no directory.  Ranking: a pair's slot inside (tile, partition) is
//   pairs of the partition in earlier waves + in earlier rounds of this wave + in lower lanes of this round.
// The middle term comes from one byte counter per (wave, partition, round) -- eight rounds = one 8-byte LDS word, filled
// with LDS atomics that return nothing, summed below a round with v_sad_u8 -- so no round waits for another one.  The
// tile's pairs are laid out partition by partition in LDS and copied out: a (tile, partition) cell is one contiguous run
// of nontemporal RowID stores.  The 131 070-element PosList cuts (one per ~32 tiles) are probe_cuts' business.
//
// LDS, in 4-byte words: staged pairs | byte counters | per-(wave, partition) pairs of earlier waves | first slot per
// partition (+ total) | output base per partition | wave totals.
__host__ __device__ constexpr size_t probe_emit_cached_lds_words(uint32_t partitions) {
  return 2 * size_t{JOIN_TILE} + 2 * size_t{JOIN_WAVES} * partitions + size_t{JOIN_WAVES} * partitions + (partitions + 2) + 2 * size_t{partitions} + 16;
}

// Sum of the byte counters of rounds 0 .. round-1 (round is a constant after unrolling).
__device__ __forceinline__ uint32_t pairs_of_earlier_rounds(u32x2_t counters, uint32_t round) {
  if (round == 0) return 0;
  if (round < 4) return __builtin_amdgcn_sad_u8(counters.x & ((1u << (8 * round)) - 1), 0u, 0u);
  if (round == 4) return __builtin_amdgcn_sad_u8(counters.x, 0u, 0u);
  return __builtin_amdgcn_sad_u8(counters.x, 0u, __builtin_amdgcn_sad_u8(counters.y & ((1u << (8 * (round - 4))) - 1), 0u, 0u));
}

struct TileLoads {   // what is loaded one tile ahead
  uint32_t meta[JOIN_ROUNDS / 2];
  uint32_t position[JOIN_ROUNDS];
  uint32_t pairs;      // thread = partition: pairs of the tile's cell
  uint32_t uncached;
};

__device__ __forceinline__ void issue_tile_loads(const ProbeArgs& a, uint32_t tile, uint32_t partitions, uint32_t tid, TileLoads& t) {
  const uint32_t* meta = a.row_meta + static_cast<size_t>(tile) * (JOIN_TILE / 2) + (tid >> 6) * (JOIN_ROUNDS / 2 * 64) + (tid & 63);
#pragma unroll
  for (uint32_t j = 0; j < JOIN_ROUNDS / 2; ++j) t.meta[j] = meta[j * 64];
  const uint32_t* position = a.row_partner + static_cast<size_t>(tile) * JOIN_TILE + (tid >> 6) * JOIN_WAVE_ROWS + (tid & 63);
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) t.position[k] = position[k * 64];   // (rows without a pair: never written, never used)
  t.pairs = a.hist_pairs[static_cast<size_t>(tid < partitions ? tid : 0) * a.n_tiles + tile];
  t.uncached = a.tile_uncached[tile];
}

__global__ __launch_bounds__(JOIN_THREADS) __attribute__((amdgpu_waves_per_eu(6))) void probe_emit_cached(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t join_smem[];
  const uint32_t partitions = 1u << a.radix_bits;
  u32x2_t* s_stage = reinterpret_cast<u32x2_t*>(join_smem);                            // [JOIN_TILE] row | partition << 12 | null << 21 , partner
  u32x2_t* s_round_pairs = s_stage + JOIN_TILE;                                        // [JOIN_WAVES][partitions] 8 byte counters, one per round
  uint32_t* s_wave_pairs = reinterpret_cast<uint32_t*>(s_round_pairs + JOIN_WAVES * partitions);   // [JOIN_WAVES][partitions] pairs of earlier waves
  uint32_t* s_tile_offset = s_wave_pairs + JOIN_WAVES * partitions;                    // [partitions + 1] first staged slot of every partition
  uint64_t* s_out_base = reinterpret_cast<uint64_t*>(s_tile_offset + partitions + 2 - (partitions & 1 ? 1 : 0));   // [partitions] global pair index of slot 0 of the partition's run, minus that slot
  uint32_t* s_scratch = reinterpret_cast<uint32_t*>(s_out_base + partitions);         // [JOIN_WAVES] wave totals of the partition scan, [2] tickets
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index in an SGPR)
  const uint32_t scan_waves = partitions > 64 ? partitions / 64 : 1;

  // this workgroup's tiles: XCD x = blockIdx % 8 owns the x-th eighth of the tiles; its workgroups draw them one at a time
  // from the XCD's ticket counter (workgroups differ in speed by a third: equal shares leave the chip half idle at the
  // end), two tickets ahead -- the next tile's loads are in flight while this one is ranked.
  const uint32_t per_xcd = (a.n_tiles + 7) / 8, xcd = blockIdx.x & 7;
  const uint32_t xcd_begin = xcd * per_xcd, xcd_end = (xcd + 1) * per_xcd < a.n_tiles ? (xcd + 1) * per_xcd : a.n_tiles;
  if (!a.plan->fits || *a.n_uncached == a.n_tiles) return;   // (every tile is probe_emit_generic's: nothing to walk)
  if (tid == 0) {
    s_scratch[JOIN_WAVES] = atomicAdd(a.xcd_tickets + xcd, 1u);
    s_scratch[JOIN_WAVES + 1] = atomicAdd(a.xcd_tickets + xcd, 1u);
  }
  __syncthreads();
  uint32_t tile = xcd_begin + s_scratch[JOIN_WAVES], tile_after = xcd_begin + s_scratch[JOIN_WAVES + 1];
  __syncthreads();
  if (tile >= xcd_end) return;
  TileLoads next;
  issue_tile_loads(a, tile, partitions, tid, next);
  for (uint32_t iteration = 0; tile < xcd_end; ++iteration) {
    // what arrived for this tile (one explicit vmcnt(0): everything in flight here is this tile's prefetch, and the waits
    // the compiler derives for single registers across the loop edge would later also cover this iteration's gathers)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // (Two things would make the wave wait for the ticket right here: an initial value to merge the atomic's result with,
    // and an address the compiler can see to be uniform -- it then aggregates the add over the wave and needs the result
    // at once to hand out the lanes' shares.  Hence the undefined start and the laundered zero offset.)
    uint32_t ticket, opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    asm volatile("" : "=v"(ticket));
    if (tid == 0) ticket = atomicAdd(a.xcd_tickets + xcd + opaque_zero, 1u);   // the tile after the next one
    uint32_t cur_meta[JOIN_ROUNDS / 2];
#pragma unroll
    for (uint32_t j = 0; j < JOIN_ROUNDS / 2; ++j) cur_meta[j] = next.meta[j];
    const uint32_t cur_pairs = next.pairs, cur_uncached = next.uncached;
    const uint32_t cur_tile = tile;
    tile = tile_after;
    // this tile's remaining loads, needed two or three barriers from here: the partners' RowIDs (d) -- fetched HERE, where
    // consecutive lanes hold consecutive probe rows (sorted probe keys: nearly consecutive build positions, a line or two
    // per wave), not in the copy-out, whose consecutive slots are rows of one partition = build positions a partition
    // count apart = one line per slot -- and the cell's base (c) ...
    uint32_t partner[JOIN_ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      const uint32_t meta = k & 1 ? cur_meta[k / 2] >> 16 : cur_meta[k / 2];
      const bool has_partner = (meta & ROW_EMIT) && !(meta & ROW_NULL_PARTNER);
      partner[k] = a.pack_build_ids ? a.dir.ids32[has_partner ? next.position[k] : 0] : next.position[k];
    }
    const uint64_t cell_base_pairs = a.base_pairs[static_cast<size_t>(tid < partitions ? tid : 0) * a.n_tiles + cur_tile];
    uint32_t chunk, row_begin, row_count;
    tile_rows(a, cur_tile, &chunk, &row_begin, &row_count);
    // ... and, behind them, the next tile's first loads (unconditionally -- after the last tile: this tile's again -- so that
    // the wait counts in front of this tile's loads can leave them in flight)
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would issue them first)
    issue_tile_loads(a, tile < xcd_end ? tile : cur_tile, partitions, tid, next);
    if (!cur_uncached) {   // (else: probe_emit_generic's tile)
    if (a.trace && tid == 0) a.trace[cur_tile * 6 + 0] = wall_clock64();
    // (a) clear the byte counters; the tile's first slot per partition from pass 1's pair counts (scan inside each wave
    //     now, across waves in (c))
    for (uint32_t i = tid; i < JOIN_WAVES * partitions; i += JOIN_THREADS) s_round_pairs[i] = u32x2_t{0, 0};
    if (wave < scan_waves) {
      const uint32_t mine = tid < partitions ? cur_pairs : 0;
      const uint32_t inclusive = join_wave_inclusive_scan(mine);
      if (tid < partitions) s_tile_offset[tid] = inclusive - mine;
      if (lane == 63) s_scratch[wave] = inclusive;
    }
    __syncthreads();
    // (b) count: one byte per (wave, partition, round)
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      const uint32_t meta = k & 1 ? cur_meta[k / 2] >> 16 : cur_meta[k / 2] & 0xFFFFu;
      if (meta & ROW_EMIT) atomicAdd(reinterpret_cast<uint32_t*>(s_round_pairs + wave * partitions + (meta & ROW_PARTITION)) + (k >> 2), 1u << (8 * (k & 3)));
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[cur_tile * 6 + 1] = wall_clock64();
    // (c) thread = partition: pairs of earlier waves, first slot, output base
    if (tid < partitions) {
      uint32_t run = 0;
#pragma unroll 1   // (unrolled, the eight LDS addresses are hoisted out of the tile loop and spilled)
      for (uint32_t w = 0; w < JOIN_WAVES; ++w) {
        const u32x2_t counters = s_round_pairs[w * partitions + tid];
        s_wave_pairs[w * partitions + tid] = run;
        run += __builtin_amdgcn_sad_u8(counters.x, 0u, 0u) + __builtin_amdgcn_sad_u8(counters.y, 0u, 0u);
      }
      uint32_t first = s_tile_offset[tid];   // (at most four waves hold partitions)
      first += (wave > 0 ? s_scratch[0] : 0u) + (wave > 1 ? s_scratch[1] : 0u) + (wave > 2 ? s_scratch[2] : 0u);
      s_tile_offset[tid] = first;
      s_out_base[tid] = cell_base_pairs - first;
      if (tid == partitions - 1) s_tile_offset[partitions] = first + cur_pairs;
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[cur_tile * 6 + 2] = wall_clock64();
    // (d) ranking: no round depends on another one
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      const uint32_t meta = k & 1 ? cur_meta[k / 2] >> 16 : cur_meta[k / 2] & 0xFFFFu;
      const uint32_t partition = meta & ROW_PARTITION;
      const bool emit = meta & ROW_EMIT;
      const uint64_t peers = match_any8(partition, emit);
      if (emit) {
        const u32x2_t counters = s_round_pairs[wave * partitions + partition];
        const uint32_t earlier_rounds = pairs_of_earlier_rounds(counters, k);
        const uint32_t lower_peers = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(peers >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(peers), 0u));
        const uint32_t slot = s_tile_offset[partition] + s_wave_pairs[wave * partitions + partition] + earlier_rounds + lower_peers;
        const uint32_t r = wave * JOIN_WAVE_ROWS + k * 64 + lane;
        s_stage[slot] = u32x2_t{r | (partition << 12) | ((meta & ROW_NULL_PARTNER) ? 1u << 21 : 0u), partner[k]};
      }
      __builtin_amdgcn_sched_barrier(0);   // one round at a time: several in flight do not fit the registers
    }
    if (a.trace && tid == 0) a.trace[cur_tile * 6 + 3] = wall_clock64();
    __syncthreads();
    if (a.trace && tid == 0) a.trace[cur_tile * 6 + 4] = wall_clock64();
    // (e) copy out: slot s of partition p is pair  base_pairs[p][tile] + (s - first slot of p)
    const uint32_t tile_pairs = s_tile_offset[partitions];
    for (uint32_t slot = tid; slot < tile_pairs; slot += JOIN_THREADS) {
      const u32x2_t record = s_stage[slot];
      const uint32_t tag = record.x, partner = record.y;
      const uint64_t pair_pos = s_out_base[(tag >> 12) & 0x1FF] + slot;
      const u32x2_t probe_id = {chunk, row_begin + (tag & 0xFFFu)};
      __builtin_nontemporal_store(probe_id, reinterpret_cast<u32x2_t*>(a.probe_out) + pair_pos);
      if (a.build_out) {
        u32x2_t build_id = {0xFFFFFFFFu, 0xFFFFFFFFu};
        if (!(tag & (1u << 21))) {
          if (a.pack_build_ids) build_id = u32x2_t{partner >> 16, partner & 0xFFFFu};
          else build_id = reinterpret_cast<const u32x2_t*>(a.dir.row_ids)[partner];
        }
        __builtin_nontemporal_store(build_id, reinterpret_cast<u32x2_t*>(a.build_out) + pair_pos);
      }
    }
    if (a.trace && tid == 0) a.trace[cur_tile * 6 + 5] = wall_clock64();
    }
    if (tid == 0) s_scratch[JOIN_WAVES + (iteration & 1)] = ticket;
    __syncthreads();   // the next tile clears what this one still reads
    tile_after = xcd_begin + s_scratch[JOIN_WAVES + (iteration & 1)];
  }
}

// The 131 070-element cuts of the tiles probe_emit_cached handles (join_hash_steps.hpp:655-660): output PosList s of
// group g (radix partition, or probe chunk without radix partitioning) starts at the pair index where the group's
// element number (s - first PosList of g) * 131 070 stands.  One wave per PosList: binary search for the cell holding that
// element in the scanned element counts, then a walk over the tile's 16-bit row words.
__global__ __launch_bounds__(64) void probe_cuts(ProbeArgs a, const uint64_t* group_first_cell, uint32_t n_groups) {
  const uint32_t slice = blockIdx.x, lane = threadIdx.x;
  if (!a.plan->fits || slice >= a.plan->n_slices) return;
  // Both searches: the wave looks at 64 evenly spaced entries per step (three steps for 15 000 entries, not fourteen).
  uint32_t lo = 0, hi = n_groups;   // last group whose first PosList is <= slice
  while (hi - lo > 1) {
    const uint32_t step = (hi - lo + 63) / 64, at = lo + lane * step;
    const uint32_t below = __popcll(__ballot(at < hi && a.partition_slice_base[at] <= slice));   // (monotone: the first `below` lanes)
    lo += (below - 1) * step;
    hi = lo + step < hi ? lo + step : hi;
  }
  const uint32_t group = lo;
  const uint64_t first_cell = group_first_cell ? group_first_cell[group] : static_cast<uint64_t>(group) * a.n_tiles;
  const uint64_t end_cell = group_first_cell ? group_first_cell[group + 1] : static_cast<uint64_t>(group + 1) * a.n_tiles;
  const uint64_t target = a.base_elements[first_cell] + static_cast<uint64_t>(slice - a.partition_slice_base[group]) * PROBE_SIZE_PER_CHUNK;
  uint64_t cell = first_cell, cell_end = end_cell;   // last cell of the group whose first element is <= target: it holds the element
  while (cell_end - cell > 1) {
    const uint64_t step = (cell_end - cell + 63) / 64, at = cell + lane * step;
    const uint32_t below = __popcll(__ballot(at < cell_end && a.base_elements[at] <= target));
    cell += (below - 1) * step;
    cell_end = cell + step < cell_end ? cell + step : cell_end;
  }
  const uint32_t tile = static_cast<uint32_t>(a.radix_bits ? cell - static_cast<uint64_t>(group) * a.n_tiles : cell);
  if (a.tile_uncached[tile]) return;   // probe_emit_generic records the cuts of its tiles
  const uint32_t partition = a.radix_bits ? group : 0;
  uint32_t cut_rank = static_cast<uint32_t>(target - a.base_elements[cell]), pairs_before = 0;
  const uint64_t lower_lanes = (1ull << lane) - 1;
  const uint32_t* meta = a.row_meta + static_cast<size_t>(tile) * (JOIN_TILE / 2) + lane;   // word (wave, j) holds rounds 2j and 2j+1 of the wave: row order
#pragma unroll 1
  for (uint32_t batch = 0; batch < JOIN_TILE / 128; batch += 16) {
    uint32_t word[16];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) word[k] = meta[(batch + k) * 64];
#pragma unroll
    for (uint32_t k = 0; k < 32; ++k) {
      const uint32_t row_word = k & 1 ? word[k / 2] >> 16 : word[k / 2] & 0xFFFFu;
      const bool member = (row_word & ROW_MATERIALISED) && (row_word & ROW_PARTITION) == partition;
      const uint64_t members = __ballot(member), emitters = __ballot(member && (row_word & ROW_EMIT));
      const uint32_t n = __popcll(members);
      if (cut_rank < n) {
        if (member && __popcll(members & lower_lanes) == cut_rank) a.slice_offsets[slice] = a.base_pairs[cell] + pairs_before + __popcll(emitters & lower_lanes);
        return;
      }
      cut_rank -= n;
      pairs_before += __popcll(emitters);
    }
  }
}

// ---- the probe passes over a rank table ------------------------------------------------------------------------------------
// Unique integer build keys (RankTable above): a probe row has at most one partner and finding it is one 8-byte load, so
// pass 2 evaluates the rows again instead of reading what pass 1 found -- pass 1 writes nothing per row (the 6 bytes per
// probe row it left behind for probe_emit_cached were 0.73 GB of the 2.1 GB a config-3 join moved), no tile is ever
// "uncached", and the 131 070-element cuts re-evaluate the one tile they fall into.
// meta[k] = emit << 10 | null_partner << 9 | partition (INVALID_PARTITION: not materialised); rank[k] = the partner's rank.
__device__ __forceinline__ void rt_evaluate_rows(const ProbeArgs& a, uint32_t chunk, uint32_t row_begin, uint32_t row_count, uint32_t wave, uint32_t lane,
                                                 uint32_t (&meta)[JOIN_ROUNDS], uint32_t (&rank)[JOIN_ROUNDS]) {
  uint32_t row[JOIN_ROUNDS];
  bool in[JOIN_ROUNDS], is_null[JOIN_ROUNDS];
  int64_t key[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    meta[k] = INVALID_PARTITION;
    rank[k] = 0;
  }
  if (row_count == 0) return;
  decode_keys<false>(a, chunk, row_begin, row_count, wave, lane, row, in, is_null, key);
  const RankTable& t = a.rank;
  bool valid[JOIN_ROUNDS], look[JOIN_ROUNDS];
  uint32_t rel[JOIN_ROUNDS], low[JOIN_ROUNDS];
  u32x2_t entry[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    valid[k] = in[k] && !(is_null[k] && !a.keep_nulls);
    const uint64_t distance = static_cast<uint64_t>(key[k]) - t.key_min;
    look[k] = valid[k] && !is_null[k] && distance <= t.range;
    rel[k] = look[k] ? static_cast<uint32_t>(distance) : 0u;
    low[k] = static_cast<uint32_t>(key[k]);   // std::hash of an integer key is the key: radix partition and Bloom index are its low bits
    entry[k] = t.entries[rel[k] >> 5];
  }
  bool found[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    const uint32_t bit = 1u << (rel[k] & 31);
    found[k] = look[k] && (entry[k].x & bit);
    rank[k] = entry[k].y + __popc(entry[k].x & (bit - 1));
  }
  // the build side's Bloom filter decides which partner-less probe rows count as materialised (join_hash_steps.hpp:354-358)
  if (a.build_bloom && !a.keep_nulls) {
    bool any = false;
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) any = any || (valid[k] && !found[k]);
    if (__any(any)) {
      uint32_t index[JOIN_ROUNDS];
      uint8_t hit[JOIN_ROUNDS];
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) index[k] = low[k] & (BLOOM_BITS - 1);
      load_rows<uint8_t>(a.build_bloom, index, hit);
#pragma unroll
      for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) valid[k] = valid[k] && !(!found[k] && hit[k] == 0);
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    if (!valid[k]) continue;
    const uint32_t partition = a.radix_bits ? low[k] & ((1u << a.radix_bits) - 1) : 0;
    bool null_partner = false;
    const uint32_t emit = pairs_of(a, is_null[k], found[k] ? 1u : 0u, &null_partner);
    meta[k] = (emit << 10) | (null_partner ? 0x200u : 0u) | partition;
  }
}

// ---- pass 1 over a rank table ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_view(const ProbeArgs& a, uint32_t tile, SliceView* view, uint32_t* row_begin, uint32_t* row_count) {
  *view = a.views[tile / (SLICE_ROWS / JOIN_TILE)];
  const uint32_t offset = (tile % (SLICE_ROWS / JOIN_TILE)) * JOIN_TILE;
  *row_begin = view->row_begin + offset;
  *row_count = view->row_count > offset ? (view->row_count - offset < JOIN_TILE ? view->row_count - offset : JOIN_TILE) : 0;
}

// The (tile, partition) counts and nothing else, one tile per workgroup (probe columns of any kind; columns of plain int32 /
// FrameOfReference segments take rt_stream_count below).  One LDS atomic per row: materialised elements in the low half of a
// 32-bit cell, pairs (at most one per row) in the high half; eight copies of the cells, chosen by the lane, because
// neighbouring rows share their key (four lineitems per order) and same-address LDS atomics serialise.
constexpr uint32_t COUNT_COPIES = 8;
__global__ __launch_bounds__(JOIN_THREADS) void rt_probe_count(ProbeArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cells[MAX_PARTITIONS * COUNT_COPIES];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t tile = block_tile(a.n_tiles);
  if (tile >= a.n_tiles) return;
  for (uint32_t i = tid; i < partitions * COUNT_COPIES; i += JOIN_THREADS) s_cells[i] = 0;
  __syncthreads();
  uint32_t chunk, row_begin, row_count;
  tile_rows(a, tile, &chunk, &row_begin, &row_count);
  uint32_t meta[JOIN_ROUNDS], rank[JOIN_ROUNDS];
  rt_evaluate_rows(a, chunk, row_begin, row_count, wave, lane, meta, rank);
#pragma unroll
  for (uint32_t round = 0; round < JOIN_ROUNDS; ++round) {
    const uint32_t partition = meta[round] & 0x1FF;
    if (partition != INVALID_PARTITION) atomicAdd(&s_cells[partition * COUNT_COPIES + (lane & (COUNT_COPIES - 1))], (meta[round] >> 10) ? 0x10001u : 1u);
  }
  __syncthreads();
  if (tid < partitions) {
    const u32x4_t low = *reinterpret_cast<const u32x4_t*>(s_cells + tid * COUNT_COPIES), high = *reinterpret_cast<const u32x4_t*>(s_cells + tid * COUNT_COPIES + 4);
    const uint32_t sum = low.x + low.y + low.z + low.w + high.x + high.y + high.z + high.w;
    a.hist_elements[static_cast<size_t>(tid) * a.n_tiles + tile] = sum & 0xFFFFu;
    a.hist_pairs[static_cast<size_t>(tid) * a.n_tiles + tile] = sum >> 16;
  }
}

// RowID of the build row of rank r.
__device__ __forceinline__ u32x2_t rank_row_id(const ProbeArgs& a, uint32_t r) {
  if (a.rank.identity_rows) {
    uint32_t chunk = static_cast<uint32_t>(static_cast<double>(r) * a.rank.identity_inverse);
    if (chunk * a.rank.identity_rows > r) --chunk;   // (the product is within one ulp of the quotient)
    uint32_t offset = r - chunk * a.rank.identity_rows;
    if (offset >= a.rank.identity_rows) { ++chunk; offset -= a.rank.identity_rows; }
    return u32x2_t{chunk, offset};
  }
  if (a.dir.ids32) { const uint32_t id = a.dir.ids32[r]; return u32x2_t{id >> 16, id & 0xFFFFu}; }
  return reinterpret_cast<const u32x2_t*>(a.dir.row_ids)[r];
}

constexpr uint32_t STAGE_INVALID = 0xFFFFFFFFu;   // tag of a staging slot that holds no pair (the spare slot of a run)
enum : int { BUILD_NONE = 0, BUILD_IDENTITY = 1, BUILD_PACKED = 2, BUILD_ROW_IDS = 3, BUILD_IDENTITY_65535 = 4 };

template <int BUILD>
__device__ __forceinline__ u32x2_t rank_row_id_as(const ProbeArgs& a, uint32_t r) {
  if constexpr (BUILD == BUILD_IDENTITY_65535) {
    // chunks of 2^16 - 1 rows (Chunk::DEFAULT_SIZE): r = q * 65536 + low = q * 65535 + (q + low) -- shifts and adds instead of a
    // double-precision reciprocal and two quarter-rate integer multiplies per RowID
    uint32_t chunk = r >> 16, offset = (r & 0xFFFFu) + chunk;   // offset < 2^17
    if (offset >= 65535u) { ++chunk; offset -= 65535u; }
    if (offset >= 65535u) { ++chunk; offset -= 65535u; }
    return u32x2_t{chunk, offset};
  } else if constexpr (BUILD == BUILD_IDENTITY) {
    uint32_t chunk = static_cast<uint32_t>(static_cast<double>(r) * a.rank.identity_inverse);
    if (chunk * a.rank.identity_rows > r) --chunk;   // (the product is within one ulp of the quotient)
    uint32_t offset = r - chunk * a.rank.identity_rows;
    if (offset >= a.rank.identity_rows) { ++chunk; offset -= a.rank.identity_rows; }
    return u32x2_t{chunk, offset};
  } else if constexpr (BUILD == BUILD_PACKED) {
    const uint32_t id = a.dir.ids32[r];
    return u32x2_t{id >> 16, id & 0xFFFFu};
  } else {
    return reinterpret_cast<const u32x2_t*>(a.dir.row_ids)[r];
  }
}

template <int BUILD>
__device__ __forceinline__ void rt_copy_out(const ProbeArgs& a, const u32x2_t* s_stage, const uint64_t* s_out_base, uint32_t reserved, uint32_t chunk, uint32_t tile_row_begin,
                                            uint32_t tid) {
  for (uint32_t slot = 2 * tid; slot < reserved; slot += 2 * JOIN_THREADS) {
    const u32x4_t records = *reinterpret_cast<const u32x4_t*>(s_stage + slot);
    const uint32_t tag0 = records.x, tag1 = slot + 1 < reserved ? records.z : STAGE_INVALID;
    const bool valid0 = tag0 != STAGE_INVALID, valid1 = tag1 != STAGE_INVALID;
    const uint32_t partition0 = (tag0 >> 12) & 0x1FF, partition1 = (tag1 >> 12) & 0x1FF;
    const u32x2_t probe0 = {chunk, tile_row_begin + (tag0 & 0xFFFu)}, probe1 = {chunk, tile_row_begin + (tag1 & 0xFFFu)};
    u32x2_t build0 = {0xFFFFFFFFu, 0xFFFFFFFFu}, build1 = {0xFFFFFFFFu, 0xFFFFFFFFu};
    if constexpr (BUILD != BUILD_NONE) {
      if (valid0 && !(tag0 & (1u << 21))) build0 = rank_row_id_as<BUILD>(a, records.y);
      if (valid1 && !(tag1 & (1u << 21))) build1 = rank_row_id_as<BUILD>(a, records.w);
    }
    if (valid0 && valid1 && partition0 == partition1) {   // both pairs of one run: its first global index has the slot's parity -> aligned
      const uint64_t pair_pos = s_out_base[partition0] + slot;
      __builtin_nontemporal_store(u32x4_t{probe0.x, probe0.y, probe1.x, probe1.y}, reinterpret_cast<u32x4_t*>(reinterpret_cast<u32x2_t*>(a.probe_out) + pair_pos));
      if constexpr (BUILD != BUILD_NONE) __builtin_nontemporal_store(u32x4_t{build0.x, build0.y, build1.x, build1.y}, reinterpret_cast<u32x4_t*>(reinterpret_cast<u32x2_t*>(a.build_out) + pair_pos));
    } else {
      if (valid0) {
        const uint64_t pair_pos = s_out_base[partition0] + slot;
        __builtin_nontemporal_store(probe0, reinterpret_cast<u32x2_t*>(a.probe_out) + pair_pos);
        if constexpr (BUILD != BUILD_NONE) __builtin_nontemporal_store(build0, reinterpret_cast<u32x2_t*>(a.build_out) + pair_pos);
      }
      if (valid1) {
        const uint64_t pair_pos = s_out_base[partition1] + slot + 1;
        __builtin_nontemporal_store(probe1, reinterpret_cast<u32x2_t*>(a.probe_out) + pair_pos);
        if constexpr (BUILD != BUILD_NONE) __builtin_nontemporal_store(build1, reinterpret_cast<u32x2_t*>(a.build_out) + pair_pos);
      }
    }
  }
}

// LDS of rt_probe_emit, in 4-byte words: staged pairs (one spare slot per partition, see below) | pairs per (wave, partition),
// then pairs of earlier waves | first slot per partition (+ total) | output base per partition | wave totals.
__host__ __device__ constexpr size_t rt_probe_emit_lds_words(uint32_t partitions) {
  return 2 * (size_t{JOIN_TILE} + partitions + 2) + size_t{JOIN_WAVES} * partitions + (partitions + 2) + 2 * size_t{partitions} + 16;
}

// Pass 2 over a rank table, persistent like pass 1 (the keys of the following tile in flight while this one is ranked and
// written).  Ranking: a pair's slot inside (tile, partition) is  pairs of earlier waves + pairs of earlier rounds of the
// wave + pairs in lower lanes of the round.  The wave keeps one running counter per partition in LDS: in every round the
// lowest lane of each group of rows with the same partition (wave-level match-any) adds the group's size with ONE returning
// atomic -- LDS operations of a wave execute in order, so what it gets back are the pairs of the earlier rounds -- and the
// group reads it from that lane.  After the rounds the counters are the wave totals; thread = partition turns them into
// pairs of earlier waves.  No counting pass, no byte counters.
// The copy-out writes 16 bytes per lane and stream: the chip retires 8-byte stores at ~5 B / clock / CU (a whole tile of
// them took 5 us even with every run sequential), 16-byte stores at twice that.  To make the two pairs of a lane one
// aligned 16-byte store, partition p's run starts at a staging slot of the parity of its first global pair index -- every
// non-empty partition reserves one spare slot for that, marked invalid.

// Does a returning LDS atomic hand the lanes of one instruction their values in lane order?  Eight waves at once, four address
// patterns each (one counter; two interleaved; 128 counters hit in runs of one to seven lanes like sorted foreign keys; a
// pseudo-random spread), several rounds on the same counters: every lane compares what it got with the count of equal
// addresses in lower lanes + earlier rounds.  failures[0] counts mismatches.
__global__ __launch_bounds__(512) void lds_atomic_order_probe(uint32_t* failures, uint32_t seed) {
  __shared__ uint32_t s_counter[8][128];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < 8 * 128; i += 512) (&s_counter[0][0])[i] = 0;
  __syncthreads();
  uint32_t wrong = 0;
  for (uint32_t pattern = 0; pattern < 4; ++pattern) {
    for (uint32_t round = 0; round < 4; ++round) {
      uint32_t address;
      if (pattern == 0) address = 0;
      else if (pattern == 1) address = lane & 1;
      else if (pattern == 2) address = ((lane + round * 64 + seed) / (1 + (seed + blockIdx.x) % 7)) & 127;
      else address = ((lane * 2654435761u + round * 40503u + seed * 97u + blockIdx.x) >> 7) & 127;
      const bool take = pattern < 2 || ((lane * 7 + round + seed) % 5) != 0;   // some lanes sit a round out
      const uint64_t peers = match_any8(address, take);
      const uint32_t lower = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(peers >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(peers), 0u));
      uint32_t got = 0;
      if (take) got = atomicAdd(&s_counter[wave][address], 1u);
      const uint32_t leader = take ? static_cast<uint32_t>(__ffsll(static_cast<long long>(peers))) - 1u : lane;
      const uint32_t base = __shfl(got, static_cast<int>(leader), 64);      // the lowest lane must have seen the counter's old value ...
      if (take && got != base + lower) ++wrong;                              // ... and every lane old value + equal addresses below it
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 8 * 128; i += 512) (&s_counter[0][0])[i] = 0;
    __syncthreads();
  }
  if (wrong) atomicAdd(failures, wrong);
}

__global__ __launch_bounds__(JOIN_THREADS) void rt_probe_emit(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t join_smem[];
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t stage_slots = JOIN_TILE + partitions + 2;
  u32x2_t* s_stage = reinterpret_cast<u32x2_t*>(join_smem);                            // [stage_slots] row | partition << 12 | null << 21 , partner's rank
  uint32_t* s_wave_pairs = reinterpret_cast<uint32_t*>(s_stage + stage_slots);         // [JOIN_WAVES][partitions] running pairs of a wave, then pairs of earlier waves
  uint32_t* s_tile_offset = s_wave_pairs + JOIN_WAVES * partitions;                    // [partitions + 1] first staged slot of every partition, total
  uint64_t* s_out_base = reinterpret_cast<uint64_t*>(s_tile_offset + partitions + 2 - (partitions & 1 ? 1 : 0));   // [partitions] global pair index of staging slot 0
  uint32_t* s_scratch = reinterpret_cast<uint32_t*>(s_out_base + partitions);         // [JOIN_WAVES] wave totals of the partition scan
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t scan_waves = partitions > 64 ? partitions / 64 : 1;
  const uint32_t tile = block_tile(a.n_tiles);
  if (tile >= a.n_tiles || !a.plan->fits) return;
  if (a.trace && tid == 0) a.trace[tile * 6 + 0] = wall_clock64();
  for (uint32_t i = tid; i < JOIN_WAVES * partitions; i += JOIN_THREADS) s_wave_pairs[i] = 0;
  uint32_t chunk, tile_row_begin, tile_row_count;
  tile_rows(a, tile, &chunk, &tile_row_begin, &tile_row_count);
  // thread = partition: the cell's pairs and its first global pair index
  const size_t cell = static_cast<size_t>(tid < partitions ? tid : 0) * a.n_tiles + tile;
  const uint32_t cell_pairs = tid < partitions ? a.hist_pairs[cell] : 0;
  const uint64_t cell_base = a.base_pairs[cell];
  uint32_t meta[JOIN_ROUNDS], rank[JOIN_ROUNDS];
  rt_evaluate_rows(a, chunk, tile_row_begin, tile_row_count, wave, lane, meta, rank);
  if (a.trace && tid == 0) a.trace[tile * 6 + 1] = wall_clock64();
  // (a) reserve pairs + 1 slots per non-empty partition: scan inside each wave now, across waves in (c)
  const uint32_t reserve = cell_pairs ? cell_pairs + 1 : 0;
  uint32_t first_in_wave = 0;
  if (wave < scan_waves) {
    const uint32_t inclusive = join_wave_inclusive_scan(reserve);
    first_in_wave = inclusive - reserve;
    if (lane == 63) s_scratch[wave] = inclusive;
  }
  __syncthreads();   // the counters are zero
  // (b) rank inside the wave; the rank moves into meta[k] bits 11..
  if (a.lane_ordered_atomics) {
    // One returning LDS atomic per pair: the LDS serves the lanes of one instruction that hit the same counter in lane order
    // (and a wave's LDS instructions in program order), so the value a lane gets back is the number of pairs of its partition in
    // lower lanes and earlier rounds -- its rank.  (Not an architectural promise: lds_atomic_order_probe checks it on the device
    // the process runs on; where it does not hold, the match-any ranking below runs.)  All rounds' atomics are in flight together.
    uint32_t before[JOIN_ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
      before[k] = 0;
      if ((meta[k] >> 10) != 0) before[k] = atomicAdd(&s_wave_pairs[wave * partitions + (meta[k] & 0xFF)], 1u);
    }
#pragma unroll
    for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) meta[k] |= before[k] << 11;
  } else {
    // match-any groups, four rounds at a time (their atomics in flight together): the lowest lane of a group adds the group's pairs
#pragma unroll
    for (uint32_t half = 0; half < JOIN_ROUNDS; half += 4) {
      uint32_t before[4], who[4];   // who: leader lane | pairs in lower lanes << 8
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t k = half + j;
        const uint32_t partition = meta[k] & 0xFF;
        const bool emit = (meta[k] >> 10) != 0;
        const uint64_t peers = match_any8(partition, emit);
        const uint32_t leader = emit ? static_cast<uint32_t>(__ffsll(static_cast<long long>(peers))) - 1u : lane;
        who[j] = leader | __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(peers >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(peers), 0u)) << 8;
        before[j] = 0;
        if (emit && leader == lane) before[j] = atomicAdd(&s_wave_pairs[wave * partitions + partition], static_cast<uint32_t>(__popcll(peers)));
      }
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) meta[half + j] |= (__shfl(before[j], static_cast<int>(who[j] & 0xFF), 64) + (who[j] >> 8)) << 11;
    }
  }
  __syncthreads();
  if (a.trace && tid == 0) a.trace[tile * 6 + 2] = wall_clock64();
  // (c) thread = partition: pairs of earlier waves, first slot (parity of the first global pair), output base
  if (tid < partitions) {
    uint32_t run = 0;
#pragma unroll
    for (uint32_t w = 0; w < JOIN_WAVES; ++w) {
      const uint32_t pairs = s_wave_pairs[w * partitions + tid];
      s_wave_pairs[w * partitions + tid] = run;
      run += pairs;
    }
    uint32_t first = first_in_wave + (wave > 0 ? s_scratch[0] : 0u) + (wave > 1 ? s_scratch[1] : 0u) + (wave > 2 ? s_scratch[2] : 0u);
    if (reserve) {
      const uint32_t shift = (first ^ static_cast<uint32_t>(cell_base)) & 1u;
      s_stage[shift ? first : first + cell_pairs].x = STAGE_INVALID;   // the spare slot
      first += shift;
    }
    s_tile_offset[tid] = first;
    s_out_base[tid] = cell_base - first;
    if (tid == 0) s_tile_offset[partitions] = s_scratch[0] + (scan_waves > 1 ? s_scratch[1] : 0u) + (scan_waves > 2 ? s_scratch[2] : 0u) + (scan_waves > 3 ? s_scratch[3] : 0u);   // every reserved slot
  }
  __syncthreads();
  if (a.trace && tid == 0) a.trace[tile * 6 + 3] = wall_clock64();
  // (d) stage: slot = first slot of the partition + pairs of earlier waves + rank inside the wave
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    if (!((meta[k] >> 10) & 1)) continue;
    const uint32_t partition = meta[k] & 0xFF;
    const uint32_t slot = s_tile_offset[partition] + s_wave_pairs[wave * partitions + partition] + (meta[k] >> 11);
    const uint32_t r = wave * JOIN_WAVE_ROWS + k * 64 + lane;
    s_stage[slot] = u32x2_t{r | (partition << 12) | ((meta[k] & 0x200u) ? 1u << 21 : 0u), rank[k]};
  }
  __syncthreads();
  if (a.trace && tid == 0) a.trace[tile * 6 + 4] = wall_clock64();
  // (e) copy out, two staging slots (= two consecutive pairs of one partition, or a run's end and the next one's spare) per lane.
  // One loop per way of turning a partner's rank into its RowID -- chosen once, outside: the identity case (rank = row number) has
  // no global load in it, so the compiler does not put an `s_waitcnt vmcnt(0)` (which also waits for every STORE in flight: loads and
  // stores share the counter on gfx9) at the top of every iteration, as it did when the three cases shared one loop.
  const uint32_t reserved = s_tile_offset[partitions];
  if (!a.build_out) rt_copy_out<BUILD_NONE>(a, s_stage, s_out_base, reserved, chunk, tile_row_begin, tid);
  else if (a.rank.identity_rows == 65535u) rt_copy_out<BUILD_IDENTITY_65535>(a, s_stage, s_out_base, reserved, chunk, tile_row_begin, tid);
  else if (a.rank.identity_rows) rt_copy_out<BUILD_IDENTITY>(a, s_stage, s_out_base, reserved, chunk, tile_row_begin, tid);
  else if (a.dir.ids32) rt_copy_out<BUILD_PACKED>(a, s_stage, s_out_base, reserved, chunk, tile_row_begin, tid);
  else rt_copy_out<BUILD_ROW_IDS>(a, s_stage, s_out_base, reserved, chunk, tile_row_begin, tid);
  if (a.trace && tid == 0) a.trace[tile * 6 + 5] = wall_clock64();
}

// ---- pass 1, one wave per tile -------------------------------------------------------------------------------------------------
// Counting needs no workgroup and no row order: a WAVE takes a tile, requests all of its stored words at once with 16-byte
// loads (eight batches of 512 rows, a lane holds eight consecutive rows of each: 8 KB in flight per wave -- with the 2-byte
// loads of the workgroup kernels a CU never had more than ~25 KB in flight and the column arrived at 1.5 TB/s), then looks
// the batches up one after the other, the next batch's table entries in flight, and counts in private LDS cells.  No
// barriers; 24+ independent streams per CU.
constexpr uint32_t STREAM_WAVES = 4;   // waves per workgroup (they share nothing but the launch)
__host__ __device__ constexpr size_t stream_count_lds_words(uint32_t partitions) { return size_t{STREAM_WAVES} * partitions * COUNT_COPIES; }

template <uint32_t WIDTH>   // bytes per stored word: FrameOfReference offsets of 1 / 2 / 4 bytes, int32 values
__device__ __forceinline__ void load_batch_words(const char* base, uint32_t first_row, u32x4_t (&words)[2]) {
  // the lane's eight consecutive words (first_row is a multiple of eight: the loads are aligned)
  // (global-address-space loads: `base` comes out of a descriptor and is a generic pointer to the compiler -- flat loads, which count as LDS
  //  traffic too, so that the first wait for an LDS atomic would wait for the column as well)
  typedef __attribute__((address_space(1))) const u32x4_t global_quad;
  typedef __attribute__((address_space(1))) const u32x2_t global_pair;
  words[0] = words[1] = u32x4_t{0, 0, 0, 0};
  if constexpr (WIDTH == 1) {
    const u32x2_t v = *(global_pair*)(base + first_row);
    words[0].x = v.x; words[0].y = v.y;
  } else if constexpr (WIDTH == 2) {
    words[0] = *(global_quad*)(base + first_row * 2u);
  } else {
    words[0] = *(global_quad*)(base + first_row * 4u);
    words[1] = *(global_quad*)(base + first_row * 4u + 16u);
  }
}
template <uint32_t WIDTH>
__device__ __forceinline__ uint32_t batch_word(const u32x4_t (&words)[2], uint32_t j) {   // j = 0..7, constant after unrolling
  if constexpr (WIDTH == 1) { const uint32_t w = j < 4 ? words[0].x : words[0].y; return (w >> (8 * (j & 3))) & 0xFFu; }
  else if constexpr (WIDTH == 2) { const uint32_t w = j < 2 ? words[0].x : j < 4 ? words[0].y : j < 6 ? words[0].z : words[0].w; return (w >> (16 * (j & 1))) & 0xFFFFu; }
  else { const u32x4_t v = j < 4 ? words[0] : words[1]; return (j & 3) == 0 ? v.x : (j & 3) == 1 ? v.y : (j & 3) == 2 ? v.z : v.w; }
}

template <uint32_t WIDTH>
__device__ __forceinline__ void count_tile_wide(const ProbeArgs& a, const SliceView& view, uint32_t row_begin, uint32_t row_count, uint32_t lane, uint32_t* cells) {
  const char* base = static_cast<const char*>(view.data);
  u32x4_t words[JOIN_WAVES][2];
  uint32_t bias[JOIN_WAVES];
#pragma unroll
  for (uint32_t b = 0; b < JOIN_WAVES; ++b) {
    const uint32_t first = b * JOIN_WAVE_ROWS + lane * 8;
    load_batch_words<WIDTH>(base, row_begin + (first < row_count ? first : 0), words[b]);
    bias[b] = view.kind == VIEW_INT32 ? 0u : static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[(row_begin + (b * JOIN_WAVE_ROWS < row_count ? b * JOIN_WAVE_ROWS : 0)) / HY_FOR_BLOCK_SIZE]);
  }
  const RankTable& table = a.rank;
  const uint32_t origin = static_cast<uint32_t>(table.key_min);
#pragma unroll
  for (uint32_t b = 0; b < JOIN_WAVES; ++b) {
    uint32_t low[JOIN_ROUNDS];
    u32x2_t entry[JOIN_ROUNDS];
    uint32_t valid = 0, look = 0;
#pragma unroll
    for (uint32_t j = 0; j < JOIN_ROUNDS; ++j) {
      const bool in = b * JOIN_WAVE_ROWS + lane * 8 + j < row_count;
      const int64_t key = static_cast<int32_t>(batch_word<WIDTH>(words[b], j) + bias[b]);
      const uint64_t distance = static_cast<uint64_t>(key) - table.key_min;
      const bool looked = in && distance <= table.range;
      low[j] = static_cast<uint32_t>(key);
      entry[j] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(table.entries) + (looked ? (static_cast<uint32_t>(distance) >> 5) * 8u : 0u));
      valid |= (in ? 1u : 0u) << j;
      look |= (looked ? 1u : 0u) << j;
    }
    uint32_t found = 0;
#pragma unroll
    for (uint32_t j = 0; j < JOIN_ROUNDS; ++j) {
      if (((look >> j) & 1) && (entry[j].x & (1u << ((low[j] - origin) & 31)))) found |= 1u << j;
    }
    if (a.build_bloom && !a.keep_nulls && __any((valid & ~found) != 0)) {   // partner-less rows: materialised only if the build side's filter has their bit
      uint32_t miss = 0;
#pragma unroll
      for (uint32_t j = 0; j < JOIN_ROUNDS; ++j) {
        if (((valid & ~found) >> j) & 1) miss |= (a.build_bloom[low[j] & (BLOOM_BITS - 1)] == 0 ? 1u : 0u) << j;
      }
      valid &= ~miss;
    }
#pragma unroll
    for (uint32_t j = 0; j < JOIN_ROUNDS; ++j) {
      if (!((valid >> j) & 1)) continue;
      const uint32_t partition = a.radix_bits ? low[j] & ((1u << a.radix_bits) - 1) : 0;
      bool null_partner = false;
      const uint32_t emit = pairs_of(a, false, (found >> j) & 1, &null_partner);
      atomicAdd(&cells[partition * COUNT_COPIES + ((lane + j) & (COUNT_COPIES - 1))], emit ? 0x10001u : 1u);
    }
  }
}

__global__ __launch_bounds__(64 * STREAM_WAVES) void rt_stream_count(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t join_smem[];
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t partitions = 1u << a.radix_bits;
  uint32_t* cells = join_smem + wave * partitions * COUNT_COPIES;   // this wave's: [partitions][COUNT_COPIES]
  // the tiles of a wave: XCD x = blockIdx % 8 owns the x-th eighth of the tiles, its waves take them interleaved
  const uint32_t per_xcd = (a.n_tiles + 7) / 8, xcd = blockIdx.x & 7;
  const uint32_t end = (xcd + 1) * per_xcd < a.n_tiles ? (xcd + 1) * per_xcd : a.n_tiles, step = (gridDim.x >> 3) * STREAM_WAVES;
  for (uint32_t i = lane; i < partitions * COUNT_COPIES; i += 64) cells[i] = 0;
#pragma unroll 1
  for (uint32_t tile = xcd * per_xcd + (blockIdx.x >> 3) * STREAM_WAVES + wave; tile < end; tile += step) {
    SliceView view;
    uint32_t row_begin, row_count;
    tile_view(a, tile, &view, &row_begin, &row_count);
    if (row_count) {
      if (view.kind == VIEW_FOR8) count_tile_wide<1>(a, view, row_begin, row_count, lane, cells);
      else if (view.kind == VIEW_FOR16) count_tile_wide<2>(a, view, row_begin, row_count, lane, cells);
      else count_tile_wide<4>(a, view, row_begin, row_count, lane, cells);
    }
    // the tile's cells leave, zeroed for the next one (LDS operations of a wave execute in order: no barrier)
    for (uint32_t partition = lane; partition < partitions; partition += 64) {
      u32x4_t* mine = reinterpret_cast<u32x4_t*>(cells + partition * COUNT_COPIES);
      const u32x4_t low = mine[0], high = mine[1];
      mine[0] = u32x4_t{0, 0, 0, 0};
      mine[1] = u32x4_t{0, 0, 0, 0};
      const uint32_t sum = low.x + low.y + low.z + low.w + high.x + high.y + high.z + high.w;
      a.hist_elements[static_cast<size_t>(partition) * a.n_tiles + tile] = sum & 0xFFFFu;
      a.hist_pairs[static_cast<size_t>(partition) * a.n_tiles + tile] = sum >> 16;
    }
  }
}

// The 131 070-element cuts over a rank table: one workgroup per output PosList finds the cell that holds the PosList's first
// element (the searches of probe_cuts), evaluates that tile once more and locates the element among its rows.
__global__ __launch_bounds__(JOIN_THREADS) void rt_probe_cuts(ProbeArgs a, const uint64_t* group_first_cell, uint32_t n_groups) {
  __shared__ uint32_t s_members[JOIN_WAVES * JOIN_ROUNDS], s_emitters[JOIN_WAVES * JOIN_ROUNDS];
  const uint32_t slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!a.plan->fits || slice >= a.plan->n_slices) return;
  uint32_t lo = 0, hi = n_groups;   // last group whose first PosList is <= slice (every wave searches: the results are uniform)
  while (hi - lo > 1) {
    const uint32_t step = (hi - lo + 63) / 64, at = lo + lane * step;
    const uint32_t below = __popcll(__ballot(at < hi && a.partition_slice_base[at] <= slice));
    lo += (below - 1) * step;
    hi = lo + step < hi ? lo + step : hi;
  }
  const uint32_t group = lo;
  const uint64_t first_cell = group_first_cell ? group_first_cell[group] : static_cast<uint64_t>(group) * a.n_tiles;
  const uint64_t end_cell = group_first_cell ? group_first_cell[group + 1] : static_cast<uint64_t>(group + 1) * a.n_tiles;
  const uint64_t target = a.base_elements[first_cell] + static_cast<uint64_t>(slice - a.partition_slice_base[group]) * PROBE_SIZE_PER_CHUNK;
  uint64_t cell = first_cell, cell_end = end_cell;   // last cell of the group whose first element is <= target: it holds the element
  while (cell_end - cell > 1) {
    const uint64_t step = (cell_end - cell + 63) / 64, at = cell + lane * step;
    const uint32_t below = __popcll(__ballot(at < cell_end && a.base_elements[at] <= target));
    cell += (below - 1) * step;
    cell_end = cell + step < cell_end ? cell + step : cell_end;
  }
  const uint32_t tile = static_cast<uint32_t>(a.radix_bits ? cell - static_cast<uint64_t>(group) * a.n_tiles : cell);
  const uint32_t partition = a.radix_bits ? group : 0;
  const uint32_t cut_rank = static_cast<uint32_t>(target - a.base_elements[cell]);
  uint32_t chunk, row_begin, row_count;
  tile_rows(a, tile, &chunk, &row_begin, &row_count);
  uint32_t meta[JOIN_ROUNDS], rank[JOIN_ROUNDS];
  rt_evaluate_rows(a, chunk, row_begin, row_count, wave, lane, meta, rank);
  uint64_t members[JOIN_ROUNDS], emitters[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    const bool member = (meta[k] & 0x1FF) == partition;   // (INVALID_PARTITION is no partition)
    members[k] = __ballot(member);
    emitters[k] = __ballot(member && (meta[k] >> 10));
    if (lane == 0) { s_members[wave * JOIN_ROUNDS + k] = __popcll(members[k]); s_emitters[wave * JOIN_ROUNDS + k] = __popcll(emitters[k]); }
  }
  __syncthreads();
  // rows come wave by wave, round by round, lane by lane: members / emitters in front of this wave's first round
  uint32_t members_before = 0, emitters_before = 0;
  for (uint32_t i = 0; i < wave * JOIN_ROUNDS; ++i) { members_before += s_members[i]; emitters_before += s_emitters[i]; }
  const uint64_t lower_lanes = (1ull << lane) - 1;
#pragma unroll
  for (uint32_t k = 0; k < JOIN_ROUNDS; ++k) {
    const bool member = (members[k] >> lane) & 1;
    if (member && members_before + __popcll(members[k] & lower_lanes) == cut_rank) a.slice_offsets[slice] = a.base_pairs[cell] + emitters_before + __popcll(emitters[k] & lower_lanes);
    members_before += __popcll(members[k]);
    emitters_before += __popcll(emitters[k]);
  }
}

// LDS of probe_emit, in 4-byte words: staged pairs | the pair counts of a wave's current round | per-(wave, partition)
// running counters | per-partition offsets inside the tile | global bases of the tile's cells.
__host__ __device__ constexpr size_t probe_emit_lds_words(uint32_t partitions) {
  return 2 * size_t{JOIN_STAGE} + JOIN_THREADS + 2 * size_t{JOIN_WAVES} * partitions + (partitions + 1) + 6 * size_t{partitions} + 8;
}

// Pass 2 for the tiles pass 1 flagged (a row with several partners: build keys with duplicates): the tile is evaluated
// again, a lane keeps the lookup results of its eight rows in registers, a wave-level match-any ranking with running
// per-(wave, partition) counters gives every row its stable rank inside (partition, tile), and the pairs are first laid
// out partition by partition in LDS and then copied out, one staging buffer of pairs at a time.  Two workgroups per CU walk
// pass 1's list of such tiles.
template <bool SECONDARY>
__global__ __launch_bounds__(JOIN_THREADS) void probe_emit_generic(ProbeArgs a) {
  const uint32_t n_listed = *a.n_uncached;
  if (n_listed == 0 || !a.plan->fits) return;
  extern __shared__ __attribute__((aligned(16))) uint32_t join_smem[];
  const uint32_t partitions = 1u << a.radix_bits;
  uint32_t* s_stage = join_smem;                                     // [JOIN_STAGE][2] row | partition << 12 | null << 21 , build position
  uint32_t* s_round_emit = s_stage + 2 * JOIN_STAGE;                 // [JOIN_WAVES][64] pairs of the rows of a wave's current round
  uint32_t* s_run_elements = s_round_emit + JOIN_THREADS;            // [JOIN_WAVES][partitions]
  uint32_t* s_run_pairs = s_run_elements + JOIN_WAVES * partitions;  // [JOIN_WAVES][partitions]
  uint32_t* s_tile_offset = s_run_pairs + JOIN_WAVES * partitions;   // [partitions + 1] first staged slot of every partition
  uint64_t* s_base_pairs = reinterpret_cast<uint64_t*>(s_tile_offset + partitions + 1 + ((partitions + 1) & 1));   // [partitions]
  uint32_t* s_cut_rank = reinterpret_cast<uint32_t*>(s_base_pairs + partitions);   // [partitions] rank (in the tile) of the element that starts a new output PosList
  uint32_t* s_cut_slice = s_cut_rank + partitions;                   // [partitions] ... and the index of that PosList
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
  for (uint32_t listed = blockIdx.x; listed < n_listed; listed += gridDim.x) {
  const uint32_t tile = a.uncached_tiles[listed];
  for (uint32_t i = tid; i < 2 * JOIN_WAVES * partitions; i += JOIN_THREADS) s_run_elements[i] = 0;
  __syncthreads();
  uint32_t chunk, row_begin, row_count;
  tile_rows(a, tile, &chunk, &row_begin, &row_count);
  // thread = partition: the cell's global bases, requested now and used after the tile has been evaluated
  uint64_t cell_base_pairs = 0, cell_first_element = 0;
  uint32_t cell_slice_base = 0;
  if (tid < partitions) {
    const size_t cell = static_cast<size_t>(tid) * a.n_tiles + tile;
    const uint32_t group = a.radix_bits ? tid : chunk;
    cell_base_pairs = a.base_pairs[cell];
    cell_first_element = a.base_elements[cell] - a.partition_element_origin[group];
    cell_slice_base = a.partition_slice_base[group];
  }

  // (a) the lookup results of the lane's rows (row  wave*512 + round*64 + lane  <->  index round)
  uint32_t row_meta[JOIN_ROUNDS], row_start[JOIN_ROUNDS], row_partners[JOIN_ROUNDS];
  evaluate_rows<SECONDARY>(a, chunk, row_begin, row_count, wave, lane, row_meta, row_start, row_partners);
#pragma unroll
  for (uint32_t round = 0; round < JOIN_ROUNDS; ++round) {
    const uint32_t partition = row_meta[round] & 0x1FF;
    if (partition != INVALID_PARTITION) {
      atomicAdd(&s_run_elements[wave * partitions + partition], 1u);
      if (row_meta[round] >> 10) atomicAdd(&s_run_pairs[wave * partitions + partition], row_meta[round] >> 10);
    }
  }
  __syncthreads();
  // (b) thread = partition: exclusive prefix over the waves, tile totals, global bases of the tile's cells
  if (tid < partitions) {
    uint32_t run_e = 0, run_p = 0;
    for (uint32_t w = 0; w < JOIN_WAVES; ++w) {
      const uint32_t e = s_run_elements[w * partitions + tid], p = s_run_pairs[w * partitions + tid];
      s_run_elements[w * partitions + tid] = run_e;
      s_run_pairs[w * partitions + tid] = run_p;
      run_e += e;
      run_p += p;
    }
    s_tile_offset[tid] = run_p;   // totals for now
    s_base_pairs[tid] = cell_base_pairs;
    // 131 070-element cuts (join_hash_steps.hpp:655-660): the one element of this cell, if any, whose index inside its
    // partition (radix_bits == 0: inside its probe chunk) is a multiple of PROBE_SIZE_PER_CHUNK
    const uint64_t next_cut = (cell_first_element + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK;
    const uint64_t cut_rank = next_cut * PROBE_SIZE_PER_CHUNK - cell_first_element;
    s_cut_rank[tid] = cut_rank < run_e ? static_cast<uint32_t>(cut_rank) : 0xFFFFFFFFu;
    s_cut_slice[tid] = cell_slice_base + static_cast<uint32_t>(next_cut);
  }
  __syncthreads();
  // (c) wave 0: exclusive prefix over the partitions -> first staged slot of every partition
  if (wave == 0) {
    const uint32_t per_lane = (partitions + 63) / 64;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < per_lane; ++i) {
      const uint32_t q = lane * per_lane + i;
      if (q < partitions) mine += s_tile_offset[q];
    }
    const uint32_t inclusive = join_wave_inclusive_scan(mine);
    uint32_t running = inclusive - mine;
    for (uint32_t i = 0; i < per_lane; ++i) {
      const uint32_t q = lane * per_lane + i;
      if (q < partitions) {
        const uint32_t total = s_tile_offset[q];
        s_tile_offset[q] = running;
        running += total;
      }
    }
    if (lane == 63) s_tile_offset[partitions] = inclusive;
  }
  __syncthreads();
  const uint32_t tile_pairs = s_tile_offset[partitions];

  // (d) stable ranking inside the wave, round by round: the first staged slot of every row's pairs
  uint32_t first_slot[JOIN_ROUNDS];
#pragma unroll
  for (uint32_t round = 0; round < JOIN_ROUNDS; ++round) {
    const uint32_t meta = row_meta[round];
    const uint32_t partition = meta & 0x1FF;
    const uint32_t emit = meta >> 10;
    const bool valid = partition != INVALID_PARTITION;
    const uint64_t peers = match_any(partition, valid, a.radix_bits);
    const uint64_t lower = peers & ((1ull << lane) - 1);
    const uint64_t many = __ballot(emit > 1), one = __ballot(emit == 1);
    if (many != 0) {   // rows with several partners: the peers' pair counts are read from LDS
      s_round_emit[wave * 64 + lane] = emit;
      __builtin_amdgcn_wave_barrier();
    }
    uint32_t pairs_before = 0, pairs_total = 0;
    first_slot[round] = 0;
    if (valid) {
      if (many == 0) {
        pairs_before = __popcll(lower & one);
        pairs_total = __popcll(peers & one);
      } else {
        uint64_t rest = peers;
#pragma unroll 1
        while (rest) {
          const uint32_t j = __ffsll(static_cast<long long>(rest)) - 1;
          rest &= rest - 1;
          const uint32_t e = s_round_emit[wave * 64 + j];
          if (j < lane) pairs_before += e;
          pairs_total += e;
        }
      }
      const uint32_t element_rank = s_run_elements[wave * partitions + partition] + __popcll(lower);
      const uint32_t pair_rank = s_run_pairs[wave * partitions + partition] + pairs_before;
      if (element_rank == s_cut_rank[partition]) a.slice_offsets[s_cut_slice[partition]] = s_base_pairs[partition] + pair_rank;
      first_slot[round] = s_tile_offset[partition] + pair_rank;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers >> lane) >> 1 == 0) {   // highest peer advances the running counters of its partition
      s_run_elements[wave * partitions + partition] += static_cast<uint32_t>(__popcll(peers));
      s_run_pairs[wave * partitions + partition] += pairs_total;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // (e) the tile's pairs leave through the staging buffer, JOIN_STAGE slots at a time (a tile of 4096 probe rows with four
  //     partners each is four windows): every window is laid out partition by partition in LDS by the rows whose pairs
  //     fall into it and copied out with consecutive lanes writing consecutive RowIDs.
#pragma unroll 1
  for (uint32_t window = 0; window < tile_pairs; window += JOIN_STAGE) {
    const uint32_t window_pairs = tile_pairs - window < JOIN_STAGE ? tile_pairs - window : JOIN_STAGE;
    if (window) __syncthreads();   // the previous window has been copied out
#pragma unroll
    for (uint32_t round = 0; round < JOIN_ROUNDS; ++round) {
      const uint32_t meta = row_meta[round];
      const uint32_t emit = meta >> 10;
      if ((meta & 0x1FF) == INVALID_PARTITION || emit == 0) continue;
      const uint32_t slot = first_slot[round];
      if (slot >= window + window_pairs || slot + emit <= window) continue;   // none of this row's pairs in the window
      const uint32_t from = slot < window ? window - slot : 0, to = slot + emit < window + window_pairs ? emit : window + window_pairs - slot;   // its pairs [from, to)
      const bool null_partner = meta & 0x200u;
      const uint32_t r = wave * JOIN_WAVE_ROWS + round * 64 + lane;
      const uint32_t tag = r | ((meta & 0x1FF) << 12) | (null_partner ? 1u << 21 : 0u);
      const uint32_t start = row_start[round];
      if constexpr (!SECONDARY) {
#pragma unroll 1
        for (uint32_t j = from; j < to; ++j) reinterpret_cast<u32x2_t*>(s_stage)[slot + j - window] = u32x2_t{tag, start + j};
      } else {
        // of the key's partners, the ones that satisfy the secondary predicates are the row's pairs (a NULL partner: one
        // pair, no build row; semi / anti joins: the predicates already decided whether the probe row is written, once)
        const bool filter = !null_partner && a.build_out != nullptr;
        const uint32_t candidates = filter ? row_partners[round] : emit;
        uint32_t j = 0;
#pragma unroll 1
        for (uint32_t t = 0; t < candidates && j < to; ++t) {
          if (filter && !satisfies_secondary(a, directory_row_id(a.dir, start + t), chunk, row_begin + r)) continue;
          if (j >= from) reinterpret_cast<u32x2_t*>(s_stage)[slot + j - window] = u32x2_t{tag, start + t};
          ++j;
        }
      }
    }
    __syncthreads();
    // copy out: slot s of partition p is pair  base_pairs[p][tile] + (s - first slot of p)
    for (uint32_t s = tid; s < window_pairs; s += JOIN_THREADS) {
      const u32x2_t record = reinterpret_cast<const u32x2_t*>(s_stage)[s];
      const uint32_t tag = record.x, position = record.y;
      const uint32_t partition = (tag >> 12) & 0x1FF;
      const uint64_t pair_pos = s_base_pairs[partition] + (window + s - s_tile_offset[partition]);
      const u32x2_t probe_id = {chunk, row_begin + (tag & 0xFFFu)};
      __builtin_nontemporal_store(probe_id, reinterpret_cast<u32x2_t*>(a.probe_out) + pair_pos);
      if (a.build_out) {
        u32x2_t build_id = {0xFFFFFFFFu, 0xFFFFFFFFu};
        if (!(tag & (1u << 21))) { const hy_row_id id = directory_row_id(a.dir, position); build_id = u32x2_t{id.chunk_id, id.chunk_offset}; }
        __builtin_nontemporal_store(build_id, reinterpret_cast<u32x2_t*>(a.build_out) + pair_pos);
      }
    }
  }
  __syncthreads();   // the next listed tile reuses the staging area and the counters
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------

static uint32_t calculate_radix_bits(uint64_t build_rows) {   // join_hash.cpp:70-114
  const double l2_cache_max_usable = 1024000 * 0.75;
  const double complete_hash_map_size = static_cast<double>(build_rows) * static_cast<double>(sizeof(uint32_t)) / 0.8;
  const double cluster_count = std::max(1.0, complete_hash_map_size / l2_cache_max_usable);
  return static_cast<uint32_t>(std::min<size_t>(8, static_cast<size_t>(std::ceil(std::log2(cluster_count)))));
}

// out[0..n) = exclusive prefix sums of in, out[n] = total.  With `second_at` (a multiple of SCAN_BLOCK): in[second_at..n)
// is a second array with its own sums (out[second_at + i]); the first array must end with at least one zero so that
// out[its length] is its total.  The block-sum buffer goes back to the pool of this thread's stream on return: whoever
// gets it next runs behind these kernels.
static hy_status exclusive_scan(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t stream, uint64_t second_at = ~0ull) {
  const uint32_t n_blocks = static_cast<uint32_t>((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  if (n_blocks == 0) {
    HY_HIP(hipMemsetAsync(out, 0, 8, stream));
    return HY_OK;
  }
  DeviceBuffer sums;
  HY_TRY(sums.alloc(8 * size_t{n_blocks}));
  const uint32_t restart = second_at == ~0ull ? 0xFFFFFFFFu : static_cast<uint32_t>(second_at / SCAN_BLOCK);
  hipLaunchKernelGGL(scan_block_sums, dim3(n_blocks), dim3(256), 0, stream, in, n, sums.as<uint64_t>());
  hipLaunchKernelGGL(scan_block_offsets, dim3(1), dim3(1024), 0, stream, sums.as<uint64_t>(), n_blocks, restart, out + n);
  hipLaunchKernelGGL(scan_blocks, dim3(n_blocks), dim3(256), 0, stream, in, n, sums.as<uint64_t>(), out);
  return HY_OK;
}


// ---- what the host learns from the device during a join ----------------------------------------------------------------
// A pinned, device-mapped block per thread: small kernels store into it, the host reads it after a stream synchronise.
// (A hipMemcpyAsync into pageable memory is a blit kernel plus a host round trip each: ~20 us of idle GPU per value.)
struct JoinMailbox {
  uint64_t key_or, first_key, last_key;   // build side
  uint64_t key_min, key_max;              // ... smallest / largest key as signed 64-bit values
  uint32_t unsorted, any_null, equal_neighbours, duplicate;   // duplicate: rank_table_mark met a key twice
  uint32_t unsorted_signed, build_unconfirmed;   // build_unconfirmed: the build column contradicted its key hint (pk_plan): nothing was written
  uint64_t n_pairs;                       // after pass 1
  uint32_t n_slices, n_uncached, fits;    // fits: the result's capacities hold n_pairs / n_slices
  uint32_t error;                         // pass 2: a probe row with >= 2^22 partners
};
static thread_local JoinMailbox* t_mailbox = nullptr;      // host address
static thread_local JoinMailbox* t_mailbox_dev = nullptr;  // device address of the same memory

// What rank_table_fill_checked found out about a build column that was filled on the strength of a hint (below).  Device memory
// behind the table's arrival counters (zeroed with them); the kernel that plans the join's output (pk_plan) compares it with the
// hint: a build column that contradicts its hint makes the plan say "does not fit" -- nothing is written -- and the host, when it
// reads the mailbox (or, HY_JOIN_ASYNC, hy_join_status), drops the hint and runs the join again.
struct BuildVerdict {
  uint64_t key_min, key_max;              // smallest / largest key as signed 64-bit values
  uint32_t unsorted_signed, equal_neighbours, outside_hint, done;
};

// The rank table and the filter of a hinted build must start out as zeros (whole words are stored, the words two waves share and the
// filter's words are OR-ed with atomics): zero_vectors in front of every fill was a launch of 6 us.  A thread keeps TWO blocks instead
// and hands them out alternately, zeroed: the LAST kernel of join n (pk_emit, bound by its 0.96 GB of stores: 4 MB more do not show; in the
// fill kernel's prologue they cost what zero_vectors did) clears -- one 16-byte store per thread of its first workgroups -- the block join
// n - 1 used, which nothing reads any more when join n's kernels run (same stream).  A block is `clean` from the moment such
// a launch is queued; `dirty_*` = what its last user may have written (table words, arrival counters, verdict; the filter's 128 KB).
struct ZeroedBlocks {
  void* table = nullptr;
  size_t table_capacity = 0;
  void* filter = nullptr;   // BLOOM_BITS / 8 bytes
  size_t dirty_table = 0, dirty_filter = 0;
  bool clean = false;
  hipStream_t stream = nullptr;
};
static thread_local ZeroedBlocks t_zeroed[2];

static void free_zeroed_blocks() {
  for (ZeroedBlocks& z : t_zeroed) {
    if (z.table) (void)hipFree(z.table);
    if (z.filter) (void)hipFree(z.filter);
    z = ZeroedBlocks{};
  }
}

// hy_shutdown: the calling thread's mailbox goes with its scratch arena and pools (the next join allocates a new one)
void release_thread_join_state() {
  if (t_mailbox) (void)hipHostFree(t_mailbox);
  t_mailbox = t_mailbox_dev = nullptr;
  free_zeroed_blocks();
}

static hy_status join_mailbox(JoinMailbox** host, JoinMailbox** device) {
  if (!t_mailbox) {
    static_assert(sizeof(JoinMailbox) <= 256, "the mailbox is a 256-byte pinned block");
    HY_HIP(hipHostMalloc(reinterpret_cast<void**>(&t_mailbox), 256, hipHostMallocMapped));
    HY_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&t_mailbox_dev), t_mailbox, 0));
  }
  std::memset(t_mailbox, 0, sizeof(JoinMailbox));
  *host = t_mailbox;
  *device = t_mailbox_dev;
  return HY_OK;
}

// The same table AND the statistics of dense_key_stats in ONE pass over the build column, for a column whose extent is already
// known: the first join over a resident column leaves its key range behind as a hint (hy_column::join_hint -- encoded segments
// are immutable, abstract_encoded_segment.hpp:12-17), later joins size and fill the table from the hint while they check
// every key against it (order, duplicates, smallest / largest key, keys outside the hinted range), and the host compares the
// verdict with the hint when the join's last kernel has finished: no second read of the build keys, no host round trip between
// the build and the probe.  A verdict that does not confirm the hint discards the join's output and runs the two-pass build.
// partials: [n_slices][4] min ^ sign | max ^ sign | flags (1 unsorted, 2 equal neighbours, 4 outside the hint) | unused.
// The keys of GROUP consecutive rows of a slice a SliceView describes (int32 values / FrameOfReference offsets of WIDTH bytes), one
// 16-byte load where the width allows.
template <uint32_t WIDTH> struct FillGroup { static constexpr uint32_t ROWS = WIDTH == 4 ? 4 : 8; };
template <uint32_t WIDTH>
__device__ __forceinline__ void load_group_words(const SliceView& view, uint32_t first, uint32_t (&word)[FillGroup<WIDTH>::ROWS]) {
  const char* base = static_cast<const char*>(view.data);
  if constexpr (WIDTH == 4) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(base + size_t{first} * 4);
    word[0] = v.x; word[1] = v.y; word[2] = v.z; word[3] = v.w;
  } else if constexpr (WIDTH == 2) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(base + size_t{first} * 2);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) word[j] = (w[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
  } else {
    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(base + first);
    const uint32_t w[2] = {v.x, v.y};
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) word[j] = (w[j / 4] >> (8 * (j & 3))) & 0xFFu;
  }
}

// Key of row `row` of the chunk a view describes (one row: the key in front of a wave's first group, or of the slice).
__device__ __forceinline__ int32_t view_key(const SliceView& view, uint32_t row) {
  if (view.kind == VIEW_INT32) return static_cast<const int32_t*>(view.data)[row];
  const uint32_t stored = view.kind == VIEW_FOR8 ? static_cast<const uint8_t*>(view.data)[row] : view.kind == VIEW_FOR16 ? static_cast<const uint16_t*>(view.data)[row] : static_cast<const uint32_t*>(view.data)[row];
  return static_cast<int32_t>(stored + static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[row / HY_FOR_BLOCK_SIZE]));
}

// One slice of a column of int32 keys (every segment one a SliceView describes, `origin` an int32 value and a multiple of 32): 32-bit arithmetic
// throughout -- the wrapped difference of two int32 values is their distance, or larger than any range.  The workgroup learns the
// extent of its keys from the keys themselves (a wave reduction each, no dependent loads in front of the slice's eight), sets up its
// run of table words in LDS, and the bits of a group's keys that share a word leave with ONE LDS atomic (sorted keys: four lineitems of
// an order, eight dbgen order keys per word).
// A slice's stored words in registers: what rank_table_fill_checked loads at the top of a slice, and what rank_table_fill_stream
// requests one slice AHEAD of the one it is working on.
template <uint32_t WIDTH>
struct FillWords {
  static constexpr uint32_t GROUP = FillGroup<WIDTH>::ROWS, GROUPS = SLICE_ROWS / 256 / GROUP;
  uint32_t word[GROUPS][GROUP];
  int32_t front[GROUPS];      // lane 0 of a wave: the key in front of the group's first row
  uint32_t bias[GROUPS];      // FrameOfReference: the minimum of the group's block (a group lies in one 2048-row block)
  int32_t key_before;         // thread 0: the key in front of the slice (the last row of the nearest earlier slice with rows) ...
  bool has_before;            // ... if there is one
};

template <uint32_t WIDTH>
__device__ __forceinline__ void load_fill_words(const MaterializeArgs& a, const SliceView& view, uint32_t slice_index, uint32_t tid, FillWords<WIDTH>& w) {
  constexpr uint32_t GROUP = FillWords<WIDTH>::GROUP, GROUPS = FillWords<WIDTH>::GROUPS;
  const uint32_t row_count = view.row_count, lane = tid & 63;
#pragma unroll
  for (uint32_t i = 0; i < GROUPS; ++i) {
    const uint32_t first = (i * 256 + tid) * GROUP;
    load_group_words<WIDTH>(view, view.row_begin + (first < row_count ? first : 0), w.word[i]);
    w.front[i] = 0;
    if (lane == 0 && first > 0 && first < row_count) w.front[i] = view_key(view, view.row_begin + first - 1);   // (the other lanes: from their neighbour)
    w.bias[i] = view.kind == VIEW_INT32 ? 0u : static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[(view.row_begin + (first < row_count ? first : 0)) / HY_FOR_BLOCK_SIZE]);
  }
  // the key in front of the slice: the last row of the nearest earlier slice with rows (one thread)
  w.key_before = 0;
  w.has_before = false;
  if (tid == 0) {
    for (uint32_t before = slice_index; before-- > 0;) {
      const SliceView earlier = a.views[before];
      if (earlier.row_count == 0) continue;
      w.key_before = view_key(earlier, earlier.row_begin + earlier.row_count - 1);
      w.has_before = true;
      break;
    }
  }
}

template <uint32_t WIDTH>
__device__ __forceinline__ void fill_checked_process(const MaterializeArgs& a, const SliceView& view, const FillWords<WIDTH>& w, uint32_t origin, uint32_t range, u32x2_entry_t* entries,
                                                     uint32_t* s_bits, uint32_t* s_base, int32_t* s_extent, uint32_t tid, int32_t* low_out, int32_t* high_out, uint32_t* flags_out) {
  constexpr uint32_t GROUP = FillGroup<WIDTH>::ROWS, GROUPS = SLICE_ROWS / 256 / GROUP;
  const uint32_t row_count = view.row_count, lane = tid & 63, wave = tid >> 6;
  const auto& word = w.word;
  int32_t front[GROUPS];
#pragma unroll
  for (uint32_t i = 0; i < GROUPS; ++i) front[i] = w.front[i];
  const int32_t key_before = w.key_before;
  const bool has_before = w.has_before;
  int32_t low = 0x7FFFFFFF, high = static_cast<int32_t>(0x80000000u);
  int32_t key[GROUPS][GROUP];
#pragma unroll
  for (uint32_t i = 0; i < GROUPS; ++i) {
    const uint32_t first = (i * 256 + tid) * GROUP;
    const uint32_t bias = w.bias[i];
#pragma unroll
    for (uint32_t e = 0; e < GROUP; ++e) {
      key[i][e] = static_cast<int32_t>(word[i][e] + bias);
      if (first + e < row_count) { low = key[i][e] < low ? key[i][e] : low; high = key[i][e] > high ? key[i][e] : high; }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const int32_t other_low = __shfl_xor(low, d, 64), other_high = __shfl_xor(high, d, 64);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
  }
  if (lane == 0) { s_extent[wave] = low; s_extent[4 + wave] = high; }
  __syncthreads();
  low = min(min(s_extent[0], s_extent[1]), min(s_extent[2], s_extent[3]));
  high = max(max(s_extent[4], s_extent[5]), max(s_extent[6], s_extent[7]));
  *low_out = low;
  *high_out = high;
  // the slice's run of table words, staged in LDS if it is short enough and inside the hinted range (keys outside are flagged and skipped)
  const uint32_t low_rel = static_cast<uint32_t>(low) - origin, high_rel = static_cast<uint32_t>(high) - origin;
  const bool staged = low_rel <= range && high_rel <= range && high_rel >= low_rel && (high_rel >> 5) - (low_rel >> 5) < CHECKED_FILL_WORDS;
  const uint32_t first_word = low_rel >> 5, span = staged ? (high_rel >> 5) - first_word + 1 : 0;
  for (uint32_t i = tid; i < span; i += 256) { s_bits[i] = 0; s_base[i] = 0; }
  __syncthreads();
  if (a.keep_nulls == 0xFFFFFFFFu) { *flags_out = 0; return; }   // (HY_JOIN_FILL_DEBUG=2: loads and extent only)
  const uint32_t first_row = static_cast<uint32_t>(a.row_base[view.chunk]) + view.row_begin;   // rank of the slice's first key
  // The Bloom filter of the build side (bit = key & 0xFFFFF, join_hash_steps.hpp:252,362) as 2^20 BITS: the table's origin is a
  // multiple of 32, so a key's bit inside its table word is its bit inside its filter word, and a table word's presence bits are
  // OR-ed into filter word (table word + origin / 32) mod 2^15 -- one atomic per table word instead of one byte store per key
  // (15 M scattered byte stores were 25 us of this kernel's 60).
  uint32_t* bloom_words = reinterpret_cast<uint32_t*>(a.bloom_out);
  const uint32_t origin_word = origin >> 5;
  uint32_t flags = 0;
#pragma unroll
  for (uint32_t i = 0; i < GROUPS; ++i) {
    const uint32_t first = (i * 256 + tid) * GROUP;
    const int32_t neighbour = __shfl_up(key[i][GROUP - 1], 1, 64);
    if (lane != 0) front[i] = neighbour;
    uint32_t run_word = 0xFFFFFFFFu, run_bits = 0, run_row = 0;
    bool run_leader = false;
    auto flush = [&]() {
      if (run_word == 0xFFFFFFFFu) return;
      if (run_word - first_word < span) {
        atomicOr(&s_bits[run_word - first_word], run_bits);
        if (run_leader) s_base[run_word - first_word] = first_row + run_row + 1u;   // (+ 1: 0 = the word's first key is not in this slice)
      } else {
        atomicOr(reinterpret_cast<uint32_t*>(entries + run_word), run_bits);
        if (run_leader) reinterpret_cast<uint32_t*>(entries + run_word)[1] = first_row + run_row;
        if (bloom_words) atomicOr(bloom_words + ((run_word + origin_word) & (BLOOM_BITS / 32 - 1)), run_bits);
      }
    };
#pragma unroll
    for (uint32_t e = 0; e < GROUP; ++e) {
      if (first + e >= row_count) continue;
      const bool at_start = first + e == 0;
      const int32_t k = key[i][e], previous = at_start ? key_before : (e == 0 ? front[i] : key[i][e - 1]);
      const bool has_previous = at_start ? has_before : true;
      if (has_previous) {
        if (previous > k) flags |= 1u;
        if (previous == k) flags |= 2u;
      }
      const uint32_t rel = static_cast<uint32_t>(k) - origin;
      if (rel > range) { flags |= 4u; continue; }
      const uint32_t table_word = rel >> 5, bit = 1u << (rel & 31);
      if (table_word == run_word) { run_bits |= bit; continue; }
      flush();
      run_word = table_word;
      run_bits = bit;
      run_row = first + e;
      run_leader = !has_previous || ((static_cast<uint32_t>(previous) - origin) >> 5) != table_word;   // no earlier row shares the word: its row number is the entry's base
    }
    flush();
  }
  *flags_out = flags;
  __syncthreads();
  for (uint32_t i = tid; i < span; i += 256) {
    const uint32_t bits = s_bits[i], base = s_base[i];
    if (bits && bloom_words) atomicOr(bloom_words + ((first_word + i + origin_word) & (BLOOM_BITS / 32 - 1)), bits);
    if (i == 0 || i + 1 == span) {   // may be shared with a neighbouring slice: add the bits; the base comes from the slice with the word's first key
      if (bits) atomicOr(reinterpret_cast<uint32_t*>(entries + first_word + i), bits);
      if (base) reinterpret_cast<uint32_t*>(entries + first_word + i)[1] = base - 1u;
    } else {
      entries[first_word + i] = u32x2_entry_t{bits, base ? base - 1u : 0u};
    }
  }
}

template <uint32_t WIDTH>
__device__ __forceinline__ void fill_checked_slice(const MaterializeArgs& a, const SliceView& view, uint32_t origin, uint32_t range, u32x2_entry_t* entries, uint32_t* s_bits, uint32_t* s_base,
                                                   int32_t* s_extent, uint32_t tid, int32_t* low_out, int32_t* high_out, uint32_t* flags_out) {
  FillWords<WIDTH> words;
  load_fill_words<WIDTH>(a, view, blockIdx.x, tid, words);
  fill_checked_process<WIDTH>(a, view, words, origin, range, entries, s_bits, s_base, s_extent, tid, low_out, high_out, flags_out);
}

// The same table as rank_table_fill_dense AND the statistics of dense_key_stats in ONE pass over the build column, for a column
// whose extent is already known: the first join over a resident column leaves its key range behind as a hint
// (hy_column::join_hint -- encoded segments are immutable, abstract_encoded_segment.hpp:12-17), later joins size and fill the table
// from the hint while they check every key against it (order, duplicates, smallest / largest key, keys outside the hinted range),
// and the kernel that plans the output (pk_plan) compares the verdict with the hint: no second read of the build keys, no host round
// trip between the build and the probe.  A verdict that does not confirm the hint writes no output and the host runs the two-pass
// build.  Only for columns of int32 keys whose segments SliceViews describe (the host checks).
// partials: [n_slices][4] min ^ sign | max ^ sign | flags (1 unsorted, 2 equal neighbours, 4 outside the hint) | unused.
__global__ __launch_bounds__(256, 5) void rank_table_fill_checked(MaterializeArgs a, uint64_t key_min, uint64_t hint_range, u32x2_entry_t* entries, uint64_t* partials, uint32_t* ticket,
                                                                  BuildVerdict* verdict) {
  __shared__ uint32_t s_bits[CHECKED_FILL_WORDS], s_base[CHECKED_FILL_WORDS];
  __shared__ int32_t s_extent[8];
  __shared__ uint64_t s_min[4], s_max[4];
  __shared__ uint32_t s_flags, s_last;
  const uint32_t tid = threadIdx.x;
  constexpr uint64_t SIGN = 1ull << 63;
  const SliceView view = a.views[blockIdx.x];
  int32_t low32 = 0x7FFFFFFF, high32 = static_cast<int32_t>(0x80000000u);
  uint32_t flags = 0;
  if (tid == 0) s_flags = 0;
  if (view.row_count != 0) {
    const uint32_t origin = static_cast<uint32_t>(key_min), range = static_cast<uint32_t>(hint_range);
    if (view.kind == VIEW_FOR16) fill_checked_slice<2>(a, view, origin, range, entries, s_bits, s_base, s_extent, tid, &low32, &high32, &flags);
    else if (view.kind == VIEW_FOR8) fill_checked_slice<1>(a, view, origin, range, entries, s_bits, s_base, s_extent, tid, &low32, &high32, &flags);
    else fill_checked_slice<4>(a, view, origin, range, entries, s_bits, s_base, s_extent, tid, &low32, &high32, &flags);
  }
  if (flags) atomicOr(&s_flags, flags);
  __syncthreads();
  if (tid == 0) {
    // The record leaves with agent-scope atomic stores (write-through, sc1) and is read by the last workgroup with agent-scope
    // atomic loads: no release fence here -- a fence writes back the XCD's whole L2, which holds megabytes of freshly written table
    // words (measured: 103 us for this kernel with a fence per workgroup, half of that without).
    const uint64_t low = view.row_count ? static_cast<uint64_t>(static_cast<int64_t>(low32)) ^ SIGN : ~0ull, high = view.row_count ? static_cast<uint64_t>(static_cast<int64_t>(high32)) ^ SIGN : 0;
    uint64_t* record = partials + 4 * size_t{blockIdx.x};
    __hip_atomic_store(record + 0, low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(record + 1, high, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(record + 2, static_cast<uint64_t>(s_flags), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Arrival in two levels -- 32 counters (one 128-byte line each), then one: agent-scope atomics on ONE word retire at ~88 per
    // microsecond, and 1 831 workgroups queueing on it were 22 us of this kernel.
    const uint32_t lanes = gridDim.x < CHECKED_FILL_TICKETS ? gridDim.x : CHECKED_FILL_TICKETS, mine = blockIdx.x % lanes;
    const uint32_t quota = gridDim.x / lanes + (mine < gridDim.x % lanes ? 1u : 0u);
    uint32_t last = 0;
    if (__hip_atomic_fetch_add(ticket + 32 * (1 + mine), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == quota)
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == lanes ? 1u : 0u;
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the workgroup that arrives last sees every slice's record: the verdict
  uint64_t low = ~0ull, high = 0, bits = 0;
  for (uint32_t i = tid; i < a.n_slices; i += 256) {
    uint64_t* record = partials + 4 * size_t{i};
    const uint64_t record_low = __hip_atomic_load(record + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), record_high = __hip_atomic_load(record + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    low = record_low < low ? record_low : low;
    high = record_high > high ? record_high : high;
    bits |= __hip_atomic_load(record + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    bits |= __shfl_xor(bits, d, 64);
    const uint64_t other_low = __shfl_xor(low, d, 64), other_high = __shfl_xor(high, d, 64);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
  }
  __syncthreads();
  if ((tid & 63) == 0) { s_min[tid >> 6] = low; s_max[tid >> 6] = high; atomicOr(&s_flags, static_cast<uint32_t>(bits)); }
  __syncthreads();
  if (tid != 0) return;
  for (uint32_t w = 0; w < 4; ++w) { low = s_min[w] < low ? s_min[w] : low; high = s_max[w] > high ? s_max[w] : high; }
  verdict->key_min = low ^ SIGN;
  verdict->key_max = high ^ SIGN;
  verdict->unsorted_signed = s_flags & 1u ? 1 : 0;
  verdict->equal_neighbours = s_flags & 2u ? 1 : 0;
  verdict->outside_hint = s_flags & 4u ? 1 : 0;
  verdict->done = 1;
  __threadfence_system();
}

// The same pass, wave by wave: rank_table_fill_checked spends ~130 vector instructions per key (runs of keys per table word found
// with per-key branches, five workgroup barriers per slice) and is bound by exactly that -- its time falls with the number of resident
// workgroups, not with the loads' latency (tools/join_bench.py: 1 / 2 / 3 persistent workgroups per CU 102 / 60 / 47 us).  Here a wave
// owns a run of consecutive 512-row steps (eight consecutive rows per lane: one or two 16-byte loads, requested a step ahead) and builds
// a step's table words in a window of LDS that belongs to it alone: a key is ONE ds_or (its bit) and ONE ds_min (its rank: the smallest
// rank of a word is its base) -- no run detection, no workgroup barrier, uniform control flow.  The window is [word of the step's first
// key, word of its last key]: sorted keys stay inside, and a step whose keys are not sorted (or span more than FW_WINDOW words: a very
// sparse stretch) takes one global atomic per key instead.  A step's first and last word may be shared with the neighbouring steps: their
// bits leave with a global atomicOr, and the base is written by the step that holds the word's first key.  The checks of
// rank_table_fill_checked (order, equal neighbours, extent, keys outside the hint) run on the same registers; a workgroup hands in one
// record.  Columns whose segments all have stored words of WIDTH bytes (hy_column::stream_width); partials: [gridDim.x][4].
constexpr uint32_t FW_STEP = 512;                        // rows per step: eight consecutive rows per lane
constexpr uint32_t FW_STEPS_PER_SLICE = SLICE_ROWS / FW_STEP;
constexpr uint32_t FW_WINDOW = 2048;                     // table words a batch (or a step) may span to be built in LDS (8 KB per wave)
static_assert(HY_FOR_BLOCK_SIZE % FW_STEP == 0, "a step lies in one FrameOfReference block");

constexpr uint32_t FW_BATCH = 4;                          // steps a wave builds in one window and requests at once, a batch ahead (2 x 8 loads of 16 bytes per lane in flight for 4-byte keys)

// What a wave carries from step to step (all of it uniform).
struct FillWaveState {
  uint64_t unsorted = 0, equal = 0, outside = 0;   // ballots: some lane met a key below / equal to its predecessor / outside the hinted range
  int32_t low = 0x7FFFFFFF, high = static_cast<int32_t>(0x80000000u);   // first and last key of the wave's rows: the extent if they are sorted
  bool any_rows = false;
  int32_t carry = 0;           // the key in front of the next step ...
  bool has_carry = false;      // ... if there is one
};

template <uint32_t WIDTH>
__device__ __forceinline__ int32_t fill_step_key(const u32x4_t& a, const u32x4_t& b, uint32_t bias, uint32_t j) {
  uint32_t stored;
  if constexpr (WIDTH == 4) stored = j < 4 ? a[j] : b[j - 4];
  else if constexpr (WIDTH == 2) stored = (a[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
  else stored = (a[j / 4] >> (8 * (j & 3))) & 0xFFu;
  return static_cast<int32_t>(stored + bias);
}

// The key in front of step `step` (the last row before it that exists), for the first step of a wave: one lane asks.
__device__ __forceinline__ bool key_in_front_of_step(const MaterializeArgs& a, uint32_t step, int32_t* key) {
  uint32_t slice = step / FW_STEPS_PER_SLICE;
  const uint32_t offset = (step % FW_STEPS_PER_SLICE) * FW_STEP;
  if (offset > 0) {   // (the slice has rows in front of the step, or the step has none itself)
    const SliceView view = a.views[slice];
    if (view.row_count > 0) { *key = view_key(view, view.row_begin + (offset <= view.row_count ? offset : view.row_count) - 1); return true; }
  }
  while (slice-- > 0) {
    const SliceView earlier = a.views[slice];
    if (earlier.row_count == 0) continue;
    *key = view_key(earlier, earlier.row_begin + earlier.row_count - 1);
    return true;
  }
  return false;
}

// The window of a batch's table words goes out: presence bits, and for every word the rank of its first key.
// A word's base is the rank of its first key: the batch's rows are consecutive and its keys ascend without repeats (anything else is
// flagged and ends the use of this table; a table for existence-only joins, which may hold repeats, is never asked for a base),
// so that rank is the batch's first rank plus the keys of the batch in the words before -- a running sum of population counts.
__device__ __forceinline__ void fill_flush_window(const uint32_t* bits_window, uint32_t span, uint32_t first_word, uint32_t origin_word, uint32_t first_rank, bool first_is_leader, uint32_t lane,
                                                  u32x2_entry_t* entries, uint32_t* bloom_words, uint32_t debug) {
  uint32_t keys_before = 0;
  for (uint32_t begin = 0; begin < span; begin += 64) {
    const uint32_t i = begin + lane;
    const uint32_t bits = i < span ? bits_window[i] : 0u;
    const uint32_t count = __popc(bits);
    uint32_t inclusive = count;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t other = __shfl_up(inclusive, d, 64);
      if (lane >= static_cast<uint32_t>(d)) inclusive += other;
    }
    const uint32_t base = first_rank + keys_before + inclusive - count;
    keys_before += __builtin_amdgcn_readlane(inclusive, 63);
    if (i >= span) continue;
    const uint32_t table_word = first_word + i;
    if (bits && bloom_words) atomicOr(bloom_words + ((table_word + origin_word) & (BLOOM_BITS / 32 - 1)), bits);
#ifdef HY_DEBUG_SWITCHES
    if (debug & 2) { if (bits == 0xDEADBEEFu && base == 0x12345678u) entries[0] = u32x2_entry_t{bits, base}; continue; }
#endif
    if (i == 0 || i + 1 == span) {   // may be shared with a neighbouring batch: add the bits; the base comes from the batch with the word's first key
      if (bits) atomicOr(reinterpret_cast<uint32_t*>(entries + table_word), bits);
      if (bits && (i != 0 || first_is_leader)) reinterpret_cast<uint32_t*>(entries + table_word)[1] = base;
    } else {
      entries[table_word] = u32x2_entry_t{bits, bits ? base : 0u};
    }
  }
}

// K consecutive steps from registers (K = 1: one step; K = FW_BATCH: a wave's batch -- one window for all of them, a quarter of the LDS
// round trips and of the shared edge words): step k has rows[k] rows (uniform; the steps with rows come first, a step with fewer than
// FW_STEP rows is the last with rows) whose first has rank first_rank[k]; the lane's eight consecutive keys of step k are a[k] / b[k] +
// bias[k].  Returns false -- nothing done -- if the keys of K > 1 steps do not fit one window (the caller then takes them step by step).
template <uint32_t WIDTH, uint32_t K>
__device__ __forceinline__ bool fill_process_steps(FillWaveState& w, const u32x4_t (&a)[K], const u32x4_t (&b)[K], const uint32_t (&bias)[K], const uint32_t (&rows)[K],
                                                   const uint32_t (&first_rank)[K], uint32_t lane, uint32_t origin, uint32_t range, u32x2_entry_t* entries, uint32_t* bloom_words,
                                                   uint32_t* bits_window, uint32_t debug) {
  // (debug: timing experiments of a -DHY_DEBUG_SWITCHES build, results are wrong then -- 2 no table stores, 4 no LDS work, 8 keys are only looked at)
  const uint32_t origin_word = origin >> 5;
  if (rows[0] == 0) return true;
  int32_t key[K][8];
  uint32_t lane_rows[K];
  int32_t first_key = 0, last_key = 0;
#pragma unroll
  for (uint32_t k = 0; k < K; ++k) {
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) key[k][j] = fill_step_key<WIDTH>(a[k], b[k], bias[k], j);
    lane_rows[k] = lane * 8 < rows[k] ? (rows[k] - lane * 8 < 8 ? rows[k] - lane * 8 : 8) : 0;   // rows of this lane (8 unless the step is the last of its slice)
    if (rows[k] != 0) {   // the step's last key (uniform): lane (rows - 1) / 8, element (rows - 1) % 8
      const uint32_t last_lane = (rows[k] - 1) >> 3, last_j = (rows[k] - 1) & 7;
      int32_t mine = key[k][0];
#pragma unroll
      for (uint32_t j = 1; j < 8; ++j) mine = last_j == j ? key[k][j] : mine;
      last_key = __builtin_amdgcn_readlane(mine, last_lane);
    }
    if (k == 0) first_key = __builtin_amdgcn_readfirstlane(key[0][0]);
  }
  // the window of table words: [word of the first key, word of the last key]
  const uint32_t first_rel = static_cast<uint32_t>(first_key) - origin, last_rel = static_cast<uint32_t>(last_key) - origin;
  const uint32_t first_word = first_rel >> 5;
  const bool windowed = first_rel <= range && last_rel <= range && last_rel >= first_rel && (last_rel >> 5) - first_word < FW_WINDOW;
  if (K > 1 && !windowed) return false;
  const uint32_t span = windowed ? (last_rel >> 5) - first_word + 1 : 0;
  if (!w.any_rows) w.low = first_key;
  w.high = last_key;
  w.any_rows = true;
  // order: every key against its predecessor (lane 0's first key of a step against the last key of the step before)
  int32_t before[K];
  bool first_has_before[K];
  {
    int32_t carry = w.carry;
    bool has_carry = w.has_carry;
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
      before[k] = __shfl_up(key[k][7], 1, 64);
      if (lane == 0) before[k] = carry;
      first_has_before[k] = lane != 0 || has_carry;
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const int32_t previous = j == 0 ? before[k] : key[k][j - 1];
        const bool checked = j < lane_rows[k] && (j != 0 || first_has_before[k]);
        w.unsorted |= __ballot(checked && key[k][j] < previous);
        w.equal |= __ballot(checked && key[k][j] == previous);
      }
      if (rows[k] != 0) {   // (a full step's last key sits in lane 63; only the last step with rows may be shorter, and nothing follows it)
        carry = __builtin_amdgcn_readlane(key[k][7], 63);
        has_carry = true;
      }
    }
  }
  // does the batch hold the first key of its first word?  (else an earlier step writes that word's base)
  const uint32_t carry_rel = static_cast<uint32_t>(w.carry) - origin;
  const bool first_is_leader = !w.has_carry || carry_rel > range || (carry_rel >> 5) != first_word;
  w.carry = last_key;
  w.has_carry = true;
  // keys outside the hinted range: sorted keys lie between the first and the last (and keys that are not sorted are flagged above)
  if (first_rel > range || last_rel > range) w.outside = 1;
#ifdef HY_DEBUG_SWITCHES
  if (debug & 8) return true;
#endif
  if (windowed) {
#ifdef HY_DEBUG_SWITCHES
    if (debug & 4) return true;
#endif
    for (uint32_t i = lane; i < span; i += 64) bits_window[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t rel = static_cast<uint32_t>(key[k][j]) - origin;
        const uint32_t at = (rel >> 5) - first_word;
        // (at >= span: the keys are not sorted -- flagged above, the join will not use this table; at < span: the word lies inside the table)
        if (j < lane_rows[k] && at < span) atomicOr(&bits_window[at], 1u << (rel & 31));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fill_flush_window(bits_window, span, first_word, origin_word, first_rank[0], first_is_leader, lane, entries, bloom_words, debug);
    __builtin_amdgcn_wave_barrier();   // (the window is read before the next batch clears it)
  } else {
    // a step outside any window (K == 1): one global atomic per key; a key that starts a table word writes the word's base
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
      const uint32_t lane_rank = first_rank[k] + lane * 8;
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t rel = static_cast<uint32_t>(key[k][j]) - origin;
        const int32_t previous = j == 0 ? before[k] : key[k][j - 1];
        const bool has_previous = j != 0 || first_has_before[k];
        w.outside |= __ballot(j < lane_rows[k] && rel > range);
        if (j < lane_rows[k] && rel <= range) {
          const uint32_t table_word = rel >> 5, bits = 1u << (rel & 31);
          atomicOr(reinterpret_cast<uint32_t*>(entries + table_word), bits);
          const uint32_t previous_rel = static_cast<uint32_t>(previous) - origin;
          if (!has_previous || previous_rel > range || (previous_rel >> 5) != table_word) reinterpret_cast<uint32_t*>(entries + table_word)[1] = lane_rank + j;
          if (bloom_words) atomicOr(bloom_words + ((table_word + origin_word) & (BLOOM_BITS / 32 - 1)), bits);
        }
      }
    }
  }
  return true;
}

// What a wave requests at once: the stored words of FW_BATCH steps and what it needs to know about them.
template <uint32_t WIDTH>
struct FillBatch {
  u32x4_t a[FW_BATCH], b[FW_BATCH];
  uint32_t bias[FW_BATCH], rows[FW_BATCH], first_rank[FW_BATCH];
};

template <uint32_t WIDTH>
__device__ __forceinline__ void load_fill_batch(const MaterializeArgs& a, uint32_t batch_step, uint32_t end_step, uint32_t chunk_rows, uint32_t lane, FillBatch<WIDTH>& batch) {
  typedef const __attribute__((address_space(1))) u32x4_t* global_x4;
  typedef const __attribute__((address_space(1))) u32x2_t* global_x2;
  // the batch's steps lie in at most two slices (FW_BATCH divides the steps of a slice: in one, in fact -- two keeps the code independent of that)
  const uint32_t slice0 = batch_step / FW_STEPS_PER_SLICE;
  const SliceView view0 = a.views[slice0];
  const SliceView view1 = a.views[slice0 + 1 < a.n_slices ? slice0 + 1 : slice0];
  // (every chunk but the last holds chunk_rows rows -- rank_table_fill_waves serves identity tables -- so a chunk's first rank needs no load)
  const uint32_t rank0 = view0.chunk * chunk_rows + view0.row_begin, rank1 = view1.chunk * chunk_rows + view1.row_begin;
#pragma unroll
  for (uint32_t k = 0; k < FW_BATCH; ++k) {
    const uint32_t step = batch_step + k;
    const bool second = step / FW_STEPS_PER_SLICE != slice0;
    const SliceView& view = second ? view1 : view0;
    const uint32_t offset = (step % FW_STEPS_PER_SLICE) * FW_STEP;
    batch.rows[k] = step < end_step && view.row_count > offset ? (view.row_count - offset < FW_STEP ? view.row_count - offset : FW_STEP) : 0;
    batch.first_rank[k] = (second ? rank1 : rank0) + offset;
    batch.a[k] = u32x4_t{0, 0, 0, 0};
    batch.b[k] = u32x4_t{0, 0, 0, 0};
    batch.bias[k] = 0;
    if (batch.rows[k] != 0) {
      const uint32_t row = view.row_begin + offset + (lane * 8 < batch.rows[k] ? lane * 8 : 0);   // (a lane without rows reads the step's first rows again)
      const char* base = static_cast<const char*>(view.data);
      if constexpr (WIDTH == 4) {
        batch.a[k] = *(global_x4)(base + size_t{row} * 4);
        batch.b[k] = *(global_x4)(base + size_t{row} * 4 + 16);
      } else if constexpr (WIDTH == 2) {
        batch.a[k] = *(global_x4)(base + size_t{row} * 2);
      } else {
        const u32x2_t v = *(global_x2)(base + row);
        batch.a[k].x = v.x;
        batch.a[k].y = v.y;
      }
      if (view.kind != VIEW_INT32) batch.bias[k] = static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[(view.row_begin + offset) / HY_FOR_BLOCK_SIZE]);
    }
  }
}

struct FillCleaning {   // two regions this launch leaves zeroed (16-byte vectors): the blocks of the join before (ZeroedBlocks)
  u32x4_t* first;
  u32x4_t* second;
  uint32_t first_vectors, second_vectors;
};

template <uint32_t WIDTH>
__global__ __launch_bounds__(256, 4) void rank_table_fill_waves(MaterializeArgs a, uint64_t key_min, uint64_t hint_range, u32x2_entry_t* entries, uint64_t* partials, uint32_t chunk_rows,
                                                             uint32_t batches_per_wave, uint32_t n_steps, uint32_t debug) {
  __shared__ uint32_t s_window[4][FW_WINDOW];   // per wave: the presence bits of the batch's table words
  __shared__ uint64_t s_min[4], s_max[4];
  __shared__ uint32_t s_flags;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr uint64_t SIGN = 1ull << 63;
  const uint32_t origin = static_cast<uint32_t>(key_min), range = static_cast<uint32_t>(hint_range);
  uint32_t* bloom_words = reinterpret_cast<uint32_t*>(a.bloom_out);
  uint32_t* bits_window = s_window[wave];
  const uint32_t steps_per_wave = batches_per_wave * FW_BATCH;
  const uint32_t first_step = (blockIdx.x * 4 + wave) * steps_per_wave;
  const uint32_t end_step = first_step + steps_per_wave < n_steps ? first_step + steps_per_wave : n_steps;
  if (tid == 0) s_flags = 0;
  FillWaveState w;
  FillBatch<WIDTH> current, ahead;
  if (first_step < end_step) {
    load_fill_batch<WIDTH>(a, first_step, end_step, chunk_rows, lane, current);
    int32_t front = 0;
    uint32_t has_front = 0;
    if (lane == 0) has_front = key_in_front_of_step(a, first_step, &front) ? 1u : 0u;
    w.carry = __builtin_amdgcn_readfirstlane(front);
    w.has_carry = __builtin_amdgcn_readfirstlane(has_front) != 0;
  }
  for (uint32_t batch_step = first_step; batch_step < end_step; batch_step += FW_BATCH) {
    if (batch_step + FW_BATCH < end_step) load_fill_batch<WIDTH>(a, batch_step + FW_BATCH, end_step, chunk_rows, lane, ahead);   // (in flight while this batch is built)
    // steps with rows come first inside a slice; a batch that straddles two slices (never, as FW_BATCH divides a slice's steps) or whose
    // keys do not fit one window goes step by step
    bool packed = true;
#pragma unroll
    for (uint32_t k = 1; k < FW_BATCH; ++k) packed = packed && (current.rows[k] == 0 || current.rows[k - 1] == FW_STEP);
    if (!packed || !fill_process_steps<WIDTH, FW_BATCH>(w, current.a, current.b, current.bias, current.rows, current.first_rank, lane, origin, range, entries, bloom_words, bits_window,
                                                        debug)) {
#pragma unroll
      for (uint32_t k = 0; k < FW_BATCH; ++k) {
        const u32x4_t one_a[1] = {current.a[k]}, one_b[1] = {current.b[k]};
        const uint32_t one_bias[1] = {current.bias[k]}, one_rows[1] = {current.rows[k]}, one_rank[1] = {current.first_rank[k]};
        fill_process_steps<WIDTH, 1>(w, one_a, one_b, one_bias, one_rows, one_rank, lane, origin, range, entries, bloom_words, bits_window, debug);
      }
    }
    current = ahead;
  }
  const uint64_t unsorted = w.unsorted, equal = w.equal, outside = w.outside;
  const int32_t low = w.low, high = w.high;
  const bool any_rows = w.any_rows;
  // the workgroup's record: extent and flags of its four waves.  No arrival counter, no verdict here: the kernel that plans the join's
  // output (pk_plan) reads the records -- rank_table_fill_checked's workgroups queue on two dependent agent-scope round trips for that
  const uint32_t flags = (unsorted ? 1u : 0u) | (equal ? 2u : 0u) | (outside ? 4u : 0u);
  if (lane == 0) {
    s_min[wave] = any_rows ? static_cast<uint64_t>(static_cast<int64_t>(low)) ^ SIGN : ~0ull;
    s_max[wave] = any_rows ? static_cast<uint64_t>(static_cast<int64_t>(high)) ^ SIGN : 0;
  }
  __syncthreads();
  if (lane == 0 && flags) atomicOr(&s_flags, flags);
  __syncthreads();
  if (tid == 0) {
    uint64_t group_low = ~0ull, group_high = 0;
    for (uint32_t v = 0; v < 4; ++v) { group_low = s_min[v] < group_low ? s_min[v] : group_low; group_high = s_max[v] > group_high ? s_max[v] : group_high; }
    uint64_t* record = partials + 4 * size_t{blockIdx.x};
    record[0] = group_low;
    record[1] = group_high;
    record[2] = s_flags;
  }
}

// flags: see check_sorted; [10] a key met twice by rank_table_mark
template <typename K>
__global__ void publish_build_flags(const uint32_t* flags, const K* keys, uint64_t n, JoinMailbox* mailbox) {
  mailbox->unsorted = flags[0];
  mailbox->equal_neighbours = flags[1];
  mailbox->any_null = flags[2];
  mailbox->duplicate = flags[10];
  mailbox->key_or = *reinterpret_cast<const uint64_t*>(flags + 4);
  mailbox->unsorted_signed = flags[3];
  mailbox->key_min = *reinterpret_cast<const uint64_t*>(flags + 6) ^ (1ull << 63);   // in signed order
  mailbox->key_max = *reinterpret_cast<const uint64_t*>(flags + 8) ^ (1ull << 63);
  mailbox->first_key = n ? key_bits_of(keys[0]) : 0;
  mailbox->last_key = n ? key_bits_of(keys[n - 1]) : 0;
  __threadfence_system();
}

// The same for a dense column read in place (dense_key_stats): first / last key = first row of the first / last row of the
// last slice that has rows.
__global__ __launch_bounds__(256) void publish_dense_flags(const uint64_t* partials, MaterializeArgs a, uint32_t first_slice, uint32_t last_slice, JoinMailbox* mailbox) {
  __shared__ uint64_t s_all[4], s_low[4], s_high[4], s_bits[4];
  const uint32_t tid = threadIdx.x;
  uint64_t all = 0, low = ~0ull, high = 0, bits = 0;
  for (uint32_t i = tid; i < a.n_slices; i += 256) {
    const uint64_t* record = partials + 4 * size_t{i};
    all |= record[0];
    low = record[1] < low ? record[1] : low;
    high = record[2] > high ? record[2] : high;
    bits |= record[3];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    all |= __shfl_xor(all, d, 64);
    bits |= __shfl_xor(bits, d, 64);
    const uint64_t other_low = __shfl_xor(low, d, 64), other_high = __shfl_xor(high, d, 64);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
  }
  if ((tid & 63) == 0) { s_all[tid >> 6] = all; s_low[tid >> 6] = low; s_high[tid >> 6] = high; s_bits[tid >> 6] = bits; }
  __syncthreads();
  if (tid != 0) return;
  for (uint32_t i = 1; i < 4; ++i) { all |= s_all[i]; low = s_low[i] < low ? s_low[i] : low; high = s_high[i] > high ? s_high[i] : high; bits |= s_bits[i]; }
  mailbox->unsorted = bits & 1 ? 1 : 0;
  mailbox->equal_neighbours = bits & 2 ? 1 : 0;
  mailbox->unsorted_signed = bits & 4 ? 1 : 0;
  mailbox->any_null = 0;
  mailbox->duplicate = 0;
  mailbox->key_or = all;
  mailbox->key_min = low ^ (1ull << 63);
  mailbox->key_max = high ^ (1ull << 63);
  const Slice first = a.slices[first_slice], last = a.slices[last_slice];
  mailbox->first_key = static_cast<uint64_t>(dense_key(a.segments[first.chunk], first.row_begin));
  mailbox->last_key = static_cast<uint64_t>(dense_key(a.segments[last.chunk], last.row_begin + last.row_count - 1));
  __threadfence_system();
}

// Between the two probe passes, on the device (the host only reads the outcome at the end of the join, or -- host-memory
// results -- before it allocates the staging buffers): where every group's elements start (group = radix partition, or
// probe chunk without radix partitioning), the first output PosList of every group (a new PosList every 131 070
// materialised probe elements of a group, join_hash_steps.hpp:655-660), the totals, whether the result buffers hold them.
__global__ __launch_bounds__(256) void plan_output(const uint64_t* base_elements, const uint64_t* base_pairs, const uint64_t* group_first_cell, uint32_t n_groups,
                                                   uint32_t n_tiles, uint64_t cells, uint64_t capacity, uint32_t slice_capacity, uint64_t* origin, uint32_t* slice_base,
                                                   uint64_t* slice_offsets, const uint32_t* n_uncached, JoinPlan* plan, JoinMailbox* mailbox) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_running;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_running = 0;
  __syncthreads();
  for (uint32_t begin = 0; begin < n_groups; begin += 256) {
    const uint32_t g = begin + tid;
    uint32_t slices = 0;
    if (g < n_groups) {
      const uint64_t first = group_first_cell ? group_first_cell[g] : static_cast<uint64_t>(g) * n_tiles;
      const uint64_t next = group_first_cell ? group_first_cell[g + 1] : static_cast<uint64_t>(g + 1) * n_tiles;
      const uint64_t from = base_elements[first], to = base_elements[next];
      origin[g] = from;
      if (g + 1 == n_groups) origin[n_groups] = to;
      slices = static_cast<uint32_t>((to - from + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK);
    }
    const uint32_t inclusive = join_wave_inclusive_scan(slices);
    if (lane == 63) s_wave[wave] = inclusive;
    __syncthreads();
    uint32_t before = s_running;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    if (g < n_groups) slice_base[g] = before + inclusive - slices;
    __syncthreads();
    if (tid == 255) s_running = before + inclusive;
    __syncthreads();
  }
  if (tid == 0) {
    const uint64_t n_pairs = base_pairs[cells];
    const uint32_t n_slices = s_running;
    const uint32_t fits = n_pairs <= capacity && n_slices <= slice_capacity ? 1u : 0u;
    if (n_groups == 0) origin[0] = 0;
    plan->fits = fits;
    plan->n_slices = n_slices;
    if (fits && slice_offsets) slice_offsets[n_slices] = n_pairs;
    mailbox->n_pairs = n_pairs;
    mailbox->n_slices = n_slices;
    mailbox->n_uncached = n_uncached ? *n_uncached : 0;
    mailbox->fits = fits;
    __threadfence_system();
  }
}

// A SliceView every lane of the workgroup reads alike, through the scalar cache (constant address space: the tables are written by
// hy_column_create, long before the kernel): as a vector load it is a round trip of its own in front of the tile's words.
__device__ __forceinline__ SliceView uniform_view(const SliceView* view) {
  typedef __attribute__((address_space(4))) const uint64_t constant_u64;
  const uint64_t address = reinterpret_cast<uint64_t>(view);
  constant_u64* q = (constant_u64*)(static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(address >> 32)))) << 32 |
                                    static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(address))));
  const uint64_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
  SliceView v;
  v.data = reinterpret_cast<const void*>(w0);
  v.aux = reinterpret_cast<const void*>(w1);
  v.chunk = static_cast<uint32_t>(w2);
  v.row_begin = static_cast<uint32_t>(w2 >> 32);
  v.row_count = static_cast<uint32_t>(w3);
  v.kind = static_cast<uint32_t>(w3 >> 32);
  return v;
}

#include "join_pkfk.hpp"

__global__ void publish_join_status(hy_join_status* status, uint64_t n_pairs, uint32_t n_slices, uint32_t fits) {
  status->n_pairs = n_pairs;
  status->n_slices = n_slices;
  status->fits = fits;
  status->build_confirmed = 1;
  status->error = 0;
  status->reserved = 0;
}

// HY_JOIN_TIMING=1: host-side wall clock at the join's synchronisation points (debug aid)
struct StageClock {
  bool on = HY_DEBUG_ENV("HY_JOIN_TIMING") != nullptr;
  std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (on) fprintf(stderr, "  join %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - start).count());
  }
};

static uint32_t device_cu_count();
static bool lds_atomics_are_lane_ordered(hipStream_t stream);

struct BuildSide {
  DeviceBuffer keys, rows, keys_tmp, rows_tmp, dir, bloom, flags, rank_entries, partials;
  FillCleaning cleaning{nullptr, nullptr, 0, 0};   // the thread's other zeroed block (ZeroedBlocks), for the join's last kernel to clear ...
  ZeroedBlocks* cleaned = nullptr;                 // ... and to mark clean when that kernel is queued
  bool hint_allows_duplicates = false;
  bool bloom_is_bits = false;    // the filter is 2^20 BITS (rank_table_fill_checked folds the table's presence words into it), not one byte per bit
  bool hinted = false;           // the rank table was filled from the column's key hint: pk_plan confirms `verdict` against the hint
  uint64_t hint_min = 0, hint_max = 0;
  const BuildVerdict* verdict = nullptr;   // (device memory, behind the table's arrival counters: rank_table_fill_checked)
  const uint64_t* fill_records = nullptr;  // ... or one record per workgroup of rank_table_fill_waves: [n_fill_records][4] smallest key | largest key (both ^ sign) | flags
  uint32_t n_fill_records = 0;
  uint64_t n = 0;
  Directory directory{};
  RankTable rank{};            // rank.entries != nullptr: unique integer keys, looked up in the rank table (directory.dir is not built)
  bool any_null = false;
};

// Materialise + (sort) + directory.  `bloom_in` (device) filters the build side (no observable effect, kept for the
// reference's element counts); `bloom_out` receives the build side's filter if wanted.
// `existence_only`: the join only asks whether a key exists (Semi / Anti without secondary predicates -- the reference's
// ExistenceOnly hash table, join_hash_steps.hpp:97-236): a sorted dense build column WITH duplicates still gets a rank table, of
// which only the presence bits mean anything.
// `bit_filter_ok`: whoever probes reads the Bloom filter as 2^20 bits (the kernels of join_pkfk.hpp) -- what the one-pass hinted build
// produces; everything else reads one byte per bit.
static hy_status prepare_build(const hy_column* build, bool keep_nulls, bool want_bloom, bool want_ids32, uint32_t hashed_type, bool allow_rank_table, bool existence_only,
                               bool bit_filter_ok, BuildSide& b, hipStream_t stream) {
  const uint32_t n_slices = build->n_slices;
  DeviceBuffer counts, offsets;
  HY_TRY(counts.alloc(4 * size_t{n_slices + 1}));
  HY_TRY(offsets.alloc(8 * size_t{n_slices + 2}));
  HY_TRY(b.flags.alloc(64));
  // (decided here, before anything is launched: a hinted build zeroes its table and the Bloom filter with one launch)
  bool dense = hashed_type == 0;   // a column whose segments cannot hold NULLs; float / double keys go through the generic decoder
  for (uint32_t c = 0; c < build->n_chunks && dense; ++c) {
    const hy_segment& seg = build->host_segments[c];
    dense = (seg.encoding == HY_ENC_UNENCODED || seg.encoding == HY_ENC_FRAME_OF_REFERENCE) && seg.nulls == nullptr;
  }
  bool uniform_chunks = dense && build->n_chunks > 0;
  for (uint32_t c = 1; c < build->n_chunks && uniform_chunks; ++c) {
    const uint32_t size = build->host_segments[c].size, first_size = build->host_segments[0].size;
    uniform_chunks = first_size > 0 && (c + 1 == build->n_chunks ? size <= first_size : size == first_size);
  }
  const bool identity_candidate = dense && build->rows && uniform_chunks && build->host_segments[0].size > 0 && allow_rank_table && option(HY_OPT_JOIN_RANK_TABLE) &&
                                  FIXED_JOIN_IDENTITY;
  bool hinted = identity_candidate && bit_filter_ok && build->join_hint.state.load(std::memory_order_acquire) == 1 &&
                (existence_only || build->join_hint.unique.load(std::memory_order_relaxed)) && option(HY_OPT_JOIN_HINT);
  for (uint32_t c = 0; c < build->n_chunks && hinted; ++c) {   // rank_table_fill_checked reads int32 keys through SliceViews, 16 bytes per load
    const hy_segment& seg = build->host_segments[c];
    hinted = reinterpret_cast<uintptr_t>(seg.data) % 16 == 0 && ((seg.encoding == HY_ENC_UNENCODED && seg.data_type == HY_TYPE_INT) || seg.encoding == HY_ENC_FRAME_OF_REFERENCE);
  }
  if (want_bloom) {
    HY_TRY(b.bloom.alloc(BLOOM_BITS));
    if (!hinted) HY_HIP(hipMemsetAsync(b.bloom.ptr, 0, BLOOM_BITS, stream));
  }
  MaterializeArgs m{};
  m.segments = build->d_segments;
  m.views = build->d_slice_views;
  m.slices = build->d_slices;
  m.n_slices = n_slices;
  m.keep_nulls = keep_nulls;
  m.bloom_in = nullptr;
  m.bloom_out = want_bloom ? b.bloom.as<uint8_t>() : nullptr;
  m.slice_counts = counts.as<uint32_t>();
  m.any_null = b.flags.as<uint32_t>() + 2;
  m.hashed_type = hashed_type;
  uint64_t total = 0;
  // A column whose segments cannot hold NULLs (value / FrameOfReference segments without a null vector) materialises
  // every row: slice offsets are row numbers, no counting pass and no host round trip.
  if (n_slices && dense) {
    m.row_base = build->d_row_base;
    total = build->rows;
  } else if (n_slices) {
    hipLaunchKernelGGL((join_materialize<0, false, false>), dim3(n_slices), dim3(256), 0, stream, m);
    hipLaunchKernelGGL(scan_counts, dim3(1), dim3(1024), 0, stream, counts.as<uint32_t>(), offsets.as<uint64_t>(), n_slices);
    HY_HIP(hipMemcpyAsync(&total, offsets.as<uint64_t>() + n_slices, 8, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
  }
  b.n = total;
  // A dense column that may be a primary key: look at it in place first (statistics), and if it is sorted, duplicate-free
  // and not too sparse fill the rank table from it -- no key array, no RowID array (a key's rank is its row number).
  if (identity_candidate && total) {
    auto identity_table = [&](u32x2_t* entries, uint64_t key_min, uint64_t key_max) {
      b.rank.entries = entries;
      b.rank.key_min = key_min;
      b.rank.range = key_max - key_min;
      b.rank.identity_rows = build->host_segments[0].size;
      b.rank.identity_inverse = 1.0 / static_cast<double>(b.rank.identity_rows);
      Directory& d = b.directory;   // nothing but the extent: the probe needs neither keys nor RowIDs
      d = Directory{};
      d.n = total;
      d.key_min = key_min;
      d.key_max = key_max;
      d.n_buckets = 1;
    };
    if (hinted) {
      const MaterializeArgs& m_in = m;
      // an earlier join over this column found a primary key in [key_min, key_max]: one pass fills the table and checks every key
      const uint64_t key_min = build->join_hint.key_min.load(std::memory_order_relaxed);
      uint64_t key_max = build->join_hint.key_max.load(std::memory_order_relaxed);
      if (option(HY_OPT_JOIN_BREAK_HINT) && key_max - key_min > 64) key_max -= 64;   // tests: a hint that does not hold
      const uint64_t origin = key_min & ~uint64_t{31};   // (the table starts at a multiple of 32: a key's bit in its table word is its bit in the Bloom filter's word)
      const uint64_t words = ((key_max - origin) >> 5) + 1;
      const size_t ticket_bytes = 128 * (size_t{CHECKED_FILL_TICKETS} + 1);
      const size_t table_bytes = 8 * (words + 2) + ticket_bytes + 64;   // the table | the arrival counters | the verdict
      const size_t table_vectors = (table_bytes + 15) / 16, bloom_vectors = want_bloom ? BLOOM_BITS / 8 / 16 : 0;   // (the filter as bits: 128 KB)
      const uint32_t stream_width = build->stream_width == 1 || build->stream_width == 2 || build->stream_width == 4 ? build->stream_width : 0;
      const bool wave_fill = stream_width && option(HY_OPT_JOIN_FILL_WGS_PER_CU) > 0;
      HY_TRY(b.partials.alloc(32 * (size_t{n_slices} * FW_STEPS_PER_SLICE / 4 + 1)));   // one record per workgroup: per slice (rank_table_fill_checked), or per four waves of >= 1 step (rank_table_fill_waves)
      // The table and the filter: one of the thread's two zeroed blocks (ZeroedBlocks), whose partner this join's pk_emit clears -- or,
      // where that does not apply (the first joins of a thread, a table that outgrew its block, the per-slice fill kernel), a block
      // zeroed by a launch of its own.
      FillCleaning cleaning{nullptr, nullptr, 0, 0};
      ZeroedBlocks* cleaned = nullptr;
      bool zeroed = false;
      if (wave_fill && FIXED_JOIN_CLEAN_TABLES && table_vectors + bloom_vectors < (1ull << 31)) {
        if ((t_zeroed[0].table && t_zeroed[0].stream != stream) || (t_zeroed[1].table && t_zeroed[1].stream != stream)) {   // another stream: start over
          HY_HIP(hipDeviceSynchronize());
          free_zeroed_blocks();
        }
        int mine = -1;
        for (int i = 0; i < 2; ++i) if (t_zeroed[i].clean && t_zeroed[i].table_capacity >= 16 * table_vectors) mine = i;
        if (mine < 0) {   // a block that is not waiting to be cleaned by this launch: (re)allocate it, zeroed by zero_vectors below
          mine = !t_zeroed[0].table ? 0 : !t_zeroed[1].table ? 1 : t_zeroed[1].clean && !t_zeroed[0].clean ? 1 : 0;   // (both in use and neither cleared -- the join before took other kernels: block 0, cleared by a launch)
          if (mine >= 0 && t_zeroed[mine].table_capacity < 16 * table_vectors) {
            ZeroedBlocks& z = t_zeroed[mine];
            if (z.table) { HY_HIP(hipStreamSynchronize(stream)); HY_HIP(hipFree(z.table)); z.table = nullptr; }
            z.table_capacity = (16 * table_vectors * 5 / 4 + 4095) & ~size_t{4095};
            HY_HIP(hipMalloc(&z.table, z.table_capacity));
            if (!z.filter) HY_HIP(hipMalloc(&z.filter, BLOOM_BITS / 8));
            z.stream = stream;
            z.clean = false;
            z.dirty_table = z.table_capacity;   // (never written: all of it is cleared once)
            z.dirty_filter = BLOOM_BITS / 8;
          }
        }
        if (mine >= 0) {
          ZeroedBlocks& z = t_zeroed[mine];
          ZeroedBlocks& other = t_zeroed[1 - mine];
          if (!z.clean) {
            hipLaunchKernelGGL(zero_vectors, dim3(static_cast<uint32_t>(std::min<size_t>(2048, (z.dirty_table / 16 + z.dirty_filter / 16 + 255) / 256))), dim3(256), 0, stream,
                               static_cast<u32x4_t*>(z.table), z.dirty_table / 16, static_cast<u32x4_t*>(z.filter), z.dirty_filter / 16);
          }
          if (other.table && !other.clean) {   // this join's pk_emit clears what the join before left in the other block
            cleaning = FillCleaning{static_cast<u32x4_t*>(other.table), static_cast<u32x4_t*>(other.filter), static_cast<uint32_t>(other.dirty_table / 16), static_cast<uint32_t>(other.dirty_filter / 16)};
            cleaned = &other;   // (clean once the fill kernel is queued, below)
          }
          z.clean = false;
          z.dirty_table = 16 * table_vectors;
          z.dirty_filter = 16 * bloom_vectors;
          b.rank_entries.borrow(z.table);
          if (want_bloom) { b.bloom.borrow(z.filter); m.bloom_out = b.bloom.as<uint8_t>(); }
          zeroed = true;
        }
      }
      if (!zeroed) {
        HY_TRY(b.rank_entries.alloc(table_bytes));
        hipLaunchKernelGGL(zero_vectors, dim3(static_cast<uint32_t>(std::min<size_t>(2048, (table_vectors + bloom_vectors + 255) / 256))), dim3(256), 0, stream,
                           b.rank_entries.as<u32x4_t>(), table_vectors, b.bloom.as<u32x4_t>(), bloom_vectors);
      }
      u32x2_t* entries = b.rank_entries.as<u32x2_t>();
      BuildVerdict* verdict = reinterpret_cast<BuildVerdict*>(reinterpret_cast<char*>(entries + words + 2) + ticket_bytes);
      MaterializeArgs m = m_in;
      if (const char* debug = HY_DEBUG_ENV("HY_JOIN_FILL_DEBUG")) {   // timing experiments only (results are wrong): 1 no filter, 2 no table either
        if (atoi(debug) & 1) m.bloom_out = nullptr;
        if (atoi(debug) >= 2 && !option(HY_OPT_JOIN_FILL_WGS_PER_CU)) m.keep_nulls = 0xFFFFFFFFu;   // (rank_table_fill_checked: loads and extent only)
      }
      hipEvent_t fill_started = nullptr, fill_stopped = nullptr;
      profile_events(&fill_started, &fill_stopped, HY_KERNEL_JOIN_BUILD);
      // rank_table_fill_waves: as many waves as stay resident (at most the option's workgroups per CU), each with the same number of consecutive batches
      uint32_t fill_waves = 0;
      if (wave_fill) {
        int per_cu = 0;
        const void* kernel = stream_width == 4 ? reinterpret_cast<const void*>(rank_table_fill_waves<4>) : stream_width == 2 ? reinterpret_cast<const void*>(rank_table_fill_waves<2>) : reinterpret_cast<const void*>(rank_table_fill_waves<1>);
        HY_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0));
        fill_waves = 4 * device_cu_count() * static_cast<uint32_t>(std::max<int64_t>(1, std::min<int64_t>(per_cu, option(HY_OPT_JOIN_FILL_WGS_PER_CU))));
      }
      uint32_t fill_debug = 0;
      if (const char* debug = HY_DEBUG_ENV("HY_JOIN_FILL_DEBUG")) fill_debug = static_cast<uint32_t>(atoi(debug));   // (wave kernel: 1 no filter, 2 no table stores, 4 no LDS work, 8 loads and checks only)
      if (fill_waves) {
        const uint32_t n_steps = n_slices * FW_STEPS_PER_SLICE, n_batches = (n_steps + FW_BATCH - 1) / FW_BATCH;
        const uint32_t batches_per_wave = (n_batches + fill_waves - 1) / fill_waves, waves = (n_batches + batches_per_wave - 1) / batches_per_wave, groups = (waves + 3) / 4;
        b.fill_records = b.partials.as<uint64_t>();
        b.n_fill_records = groups;
        if (stream_width == 4) hipExtLaunchKernelGGL(rank_table_fill_waves<4>, dim3(groups), dim3(256), 0, stream, fill_started, fill_stopped, 0, m, origin, key_max - origin, entries, b.partials.as<uint64_t>(), build->host_segments[0].size, batches_per_wave, n_steps, fill_debug);
        else if (stream_width == 2) hipExtLaunchKernelGGL(rank_table_fill_waves<2>, dim3(groups), dim3(256), 0, stream, fill_started, fill_stopped, 0, m, origin, key_max - origin, entries, b.partials.as<uint64_t>(), build->host_segments[0].size, batches_per_wave, n_steps, fill_debug);
        else hipExtLaunchKernelGGL(rank_table_fill_waves<1>, dim3(groups), dim3(256), 0, stream, fill_started, fill_stopped, 0, m, origin, key_max - origin, entries, b.partials.as<uint64_t>(), build->host_segments[0].size, batches_per_wave, n_steps, fill_debug);
        b.cleaning = cleaning;   // (pk_emit clears the other block: run_join_once marks it clean when that launch is queued)
        b.cleaned = cleaned;
      } else
      hipExtLaunchKernelGGL(rank_table_fill_checked, dim3(n_slices), dim3(256), 0, stream, fill_started, fill_stopped, 0, m, origin, key_max - origin, entries,
                            b.partials.as<uint64_t>(), reinterpret_cast<uint32_t*>(entries + words + 2), verdict);
      b.verdict = b.n_fill_records ? nullptr : verdict;
      identity_table(entries, origin, key_max);
      b.directory.key_min = key_min;
      b.bloom_is_bits = want_bloom;
      b.hinted = true;
      b.hint_allows_duplicates = existence_only;
      b.hint_min = key_min;
      b.hint_max = key_max;
      return HY_OK;
    }
    uint32_t first_slice = 0, last_slice = 0;   // slices with rows (the host knows the chunk sizes: slices are per chunk, in order)
    {
      std::vector<uint32_t> rows_of_slice;
      for (uint32_t c = 0; c < build->n_chunks; ++c) {
        const uint32_t size = build->host_segments[c].size;
        const uint32_t chunk_slices = (size + SLICE_ROWS - 1) / SLICE_ROWS;
        for (uint32_t i = 0; i < (chunk_slices ? chunk_slices : 1); ++i) rows_of_slice.push_back(size > i * SLICE_ROWS ? std::min(SLICE_ROWS, size - i * SLICE_ROWS) : 0);
      }
      if (rows_of_slice.size() != n_slices) return fail(HY_ERR_DEVICE, "join: slice table of the build column is inconsistent");
      first_slice = n_slices;
      for (uint32_t i = 0; i < n_slices; ++i) if (rows_of_slice[i]) { if (first_slice == n_slices) first_slice = i; last_slice = i; }
    }
    JoinMailbox* mailbox = nullptr;
    JoinMailbox* mailbox_dev = nullptr;
    HY_TRY(join_mailbox(&mailbox, &mailbox_dev));
    DeviceBuffer partials;
    HY_TRY(partials.alloc(32 * size_t{n_slices}));
    hipLaunchKernelGGL(dense_key_stats, dim3(n_slices), dim3(256), 0, stream, m, partials.as<uint64_t>());
    hipLaunchKernelGGL(publish_dense_flags, dim3(1), dim3(256), 0, stream, partials.as<uint64_t>(), m, first_slice, last_slice, mailbox_dev);
    HY_HIP(hipStreamSynchronize(stream));
    const uint64_t key_min = mailbox->key_min, range = mailbox->key_max - mailbox->key_min;
    const uint64_t words = (range >> 5) + 1;
    if (HY_DEBUG_ENV("HY_JOIN_TIMING")) fprintf(stderr, "  dense stats: min %lld max %lld unsorted %u signed %u equal %u total %llu\n", (long long)mailbox->key_min, (long long)mailbox->key_max, mailbox->unsorted, mailbox->unsorted_signed, mailbox->equal_neighbours, (unsigned long long)total);
    if (!mailbox->unsorted_signed && (!mailbox->equal_neighbours || existence_only) && range < 0xFFFFFF00ull && (words <= 2 * total + 4096 || words <= 65536)) {
      HY_TRY(b.rank_entries.alloc(8 * (words + 1)));
      u32x2_t* entries = b.rank_entries.as<u32x2_t>();
      HY_HIP(hipMemsetAsync(entries, 0, 8 * (words + 1), stream));
      hipLaunchKernelGGL(rank_table_fill_dense, dim3(n_slices), dim3(256), 0, stream, m, key_min, entries);
      identity_table(entries, mailbox->key_min, mailbox->key_max);
      if (build->join_hint.state.load(std::memory_order_acquire) == 0) {   // the next join over this column fills its table in one pass
        build->join_hint.key_min.store(mailbox->key_min, std::memory_order_relaxed);
        build->join_hint.key_max.store(mailbox->key_max, std::memory_order_relaxed);
        build->join_hint.unique.store(mailbox->equal_neighbours ? 0u : 1u, std::memory_order_relaxed);
        build->join_hint.state.store(1, std::memory_order_release);
      }
      return HY_OK;
    }
    // (not a primary key after all: materialise as usual)
  }
  HY_HIP(hipMemsetAsync(b.flags.ptr, 0, 64, stream));
  HY_HIP(hipMemsetAsync(b.flags.as<uint32_t>() + 6, 0xFF, 8, stream));   // the running minimum
  const bool key32 = build->data_type == HY_TYPE_INT && hashed_type == 0, id32 = want_ids32;
  const size_t key_bytes = key32 ? 4 : 8, row_bytes = id32 ? 4 : 8;
  uint64_t first_key = 0, last_key = 0;
  bool keys_were_sorted = false;
  HY_TRY(b.keys.alloc(key_bytes * (total + 4)));   // (32-bit keys: four entries of padding for the probe's 16-byte loads)
  HY_TRY(b.rows.alloc(row_bytes * total));
  // one instantiation per (key width, RowID width)
#define HY_JOIN_BUILD_KERNEL(KERNEL, GRID, BLOCK, ...)                                                                  \
  do {                                                                                                                  \
    if (key32 && id32) hipLaunchKernelGGL((KERNEL<true, true>), GRID, BLOCK, 0, stream, __VA_ARGS__);                 \
    else if (key32) hipLaunchKernelGGL((KERNEL<true, false>), GRID, BLOCK, 0, stream, __VA_ARGS__);                   \
    else if (id32) hipLaunchKernelGGL((KERNEL<false, true>), GRID, BLOCK, 0, stream, __VA_ARGS__);                    \
    else hipLaunchKernelGGL((KERNEL<false, false>), GRID, BLOCK, 0, stream, __VA_ARGS__);                             \
  } while (0)
  if (total) {
    m.slice_offsets = offsets.as<uint64_t>();
    m.keys = b.keys.ptr;
    m.row_ids = b.rows.ptr;
    if (dense) {
      HY_JOIN_BUILD_KERNEL(join_materialize_dense, dim3(n_slices), dim3(256), m);
    } else if (key32 && id32) {
      hipLaunchKernelGGL((join_materialize<1, true, true>), dim3(n_slices), dim3(256), 0, stream, m);
    } else if (key32) {
      hipLaunchKernelGGL((join_materialize<1, true, false>), dim3(n_slices), dim3(256), 0, stream, m);
    } else if (id32) {
      hipLaunchKernelGGL((join_materialize<1, false, true>), dim3(n_slices), dim3(256), 0, stream, m);
    } else {
      hipLaunchKernelGGL((join_materialize<1, false, false>), dim3(n_slices), dim3(256), 0, stream, m);
    }
    if (key32) HY_HIP(hipMemsetAsync(b.keys.as<uint32_t>() + total, 0, 16, stream));
    JoinMailbox* mailbox = nullptr;
    JoinMailbox* mailbox_dev = nullptr;
    HY_TRY(join_mailbox(&mailbox, &mailbox_dev));
    const dim3 check_grid(static_cast<uint32_t>(std::min<uint64_t>((total + 1023) / 1024, 1024)));
    if (key32) {
      hipLaunchKernelGGL(check_sorted<uint32_t>, check_grid, dim3(1024), 0, stream, b.keys.as<uint32_t>(), total, b.flags.as<uint32_t>());
      hipLaunchKernelGGL(publish_build_flags<uint32_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint32_t>(), total, mailbox_dev);
    } else {
      hipLaunchKernelGGL(check_sorted<uint64_t>, check_grid, dim3(1024), 0, stream, b.keys.as<uint64_t>(), total, b.flags.as<uint32_t>());
      hipLaunchKernelGGL(publish_build_flags<uint64_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint64_t>(), total, mailbox_dev);
    }
    HY_HIP(hipStreamSynchronize(stream));
    b.any_null = mailbox->any_null != 0;
    keys_were_sorted = mailbox->unsorted == 0;
    first_key = mailbox->first_key;   // min / max if the keys are already sorted
    last_key = mailbox->last_key;
    // Unique integer keys that are not too sparse: the rank table (struct RankTable) instead of sort + directory.
    bool rank_table = false;
    {
      const uint64_t key_min = mailbox->key_min, range = mailbox->key_max - mailbox->key_min;
      const uint64_t words = (range >> 5) + 1;
      if (allow_rank_table && hashed_type == 0 && !b.any_null && !mailbox->equal_neighbours && range < 0xFFFFFF00ull && (words <= 2 * total + 4096 || words <= 65536) &&
          option(HY_OPT_JOIN_RANK_TABLE) && !build->join_hint.has_duplicates.load(std::memory_order_relaxed)) {   // (a column in which an earlier join met a key twice: no second try)
        HY_TRY(b.rank_entries.alloc(8 * (words + 1)));
        u32x2_t* entries = b.rank_entries.as<u32x2_t>();
        HY_HIP(hipMemsetAsync(entries, 0, 8 * (words + 1), stream));
        const dim3 key_grid(static_cast<uint32_t>((total + 255) / 256));
        const bool sorted_signed = mailbox->unsorted_signed == 0;
        if (sorted_signed) {
          if (key32) hipLaunchKernelGGL(rank_table_fill_sorted<uint32_t>, key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), total, key_min, entries);
          else hipLaunchKernelGGL(rank_table_fill_sorted<uint64_t>, key_grid, dim3(256), 0, stream, b.keys.as<uint64_t>(), total, key_min, entries);
          rank_table = true;
        } else if (key32 && id32 && range < 0xFFFFFFF0ull && total >= 65536 && lds_atomics_are_lane_ordered(stream)) {
          // Unsorted 32-bit keys, many of them: SORT the (key - smallest key, RowID) pairs (the LSD radix sort below: tiles staged in LDS, runs
          // instead of scattered stores) and fill the table from the sorted keys -- the RowIDs then already stand in rank order.  Marking
          // 15 M shuffled keys with device-scope atomicOr and scattering their RowIDs to their ranks took 0.56 + 0.41 ms.
          uint32_t key_bits = 1;
          while (key_bits < 32 && (range >> key_bits) != 0) ++key_bits;
          HY_TRY(b.keys_tmp.alloc(key_bytes * (total + 4)));
          HY_TRY(b.rows_tmp.alloc(row_bytes * total));
          // (read BEFORE anything below is queued: publish_build_flags writes these pinned fields again, with distances instead of keys)
          const uint64_t remembered_min = mailbox->key_min, remembered_max = mailbox->key_max, remembered_or = mailbox->key_or;
          hipLaunchKernelGGL(shift_keys, key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), total, 0u - static_cast<uint32_t>(key_min));
          uint32_t* sorted_keys = b.keys.as<uint32_t>();
          uint32_t* sorted_rows = b.rows.as<uint32_t>();
          HY_TRY(sort_pairs_u32(&sorted_keys, &sorted_rows, b.keys_tmp.as<uint32_t>(), b.rows_tmp.as<uint32_t>(), total, key_bits, stream));
          if (sorted_keys != b.keys.as<uint32_t>()) {
            std::swap(b.keys.ptr, b.keys_tmp.ptr);
            std::swap(b.keys.capacity, b.keys_tmp.capacity);
            std::swap(b.rows.ptr, b.rows_tmp.ptr);
            std::swap(b.rows.capacity, b.rows_tmp.capacity);
          }
          HY_HIP(hipMemsetAsync(b.flags.ptr, 0, 64, stream));
          hipLaunchKernelGGL(check_sorted<uint32_t>, check_grid, dim3(1024), 0, stream, b.keys.as<uint32_t>(), total, b.flags.as<uint32_t>());
          hipLaunchKernelGGL(publish_build_flags<uint32_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint32_t>(), total, mailbox_dev);
          HY_HIP(hipStreamSynchronize(stream));
          const bool duplicates = mailbox->equal_neighbours != 0;
          mailbox->key_min = remembered_min;   // (the second look saw distances, not keys)
          mailbox->key_max = remembered_max;
          mailbox->key_or = remembered_or;
          if (duplicates) {   // not unique after all: the keys again as they were, for the directory below -- which wants them in the order of
            // their sign-extended bits (negative keys last): sorted already if there is no negative key, else its own sort runs (stable: equal
            // keys keep the build-row order they have now)
            build->join_hint.has_duplicates.store(1, std::memory_order_relaxed);
            hipLaunchKernelGGL(shift_keys, key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), total, static_cast<uint32_t>(key_min));
            HY_HIP(hipMemsetAsync(b.keys.as<uint32_t>() + total, 0, 16, stream));
            mailbox->unsorted = static_cast<int64_t>(key_min) < 0 ? 1 : 0;
            mailbox->equal_neighbours = 1;
          } else {
            hipLaunchKernelGGL(rank_table_fill_sorted<uint32_t>, key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), total, uint64_t{0}, entries);
            rank_table = true;
          }
        } else {
          if (key32) {
            hipLaunchKernelGGL(rank_table_mark<uint32_t>, key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), total, key_min, entries, b.flags.as<uint32_t>() + 10);
            hipLaunchKernelGGL(publish_build_flags<uint32_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint32_t>(), total, mailbox_dev);
          } else {
            hipLaunchKernelGGL(rank_table_mark<uint64_t>, key_grid, dim3(256), 0, stream, b.keys.as<uint64_t>(), total, key_min, entries, b.flags.as<uint32_t>() + 10);
            hipLaunchKernelGGL(publish_build_flags<uint64_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint64_t>(), total, mailbox_dev);
          }
          HY_HIP(hipStreamSynchronize(stream));
          if (mailbox->duplicate) build->join_hint.has_duplicates.store(1, std::memory_order_relaxed);
          if (!mailbox->duplicate) {   // bases by a scan over the words' population counts, then every RowID goes to its key's rank
            const uint32_t n_blocks = static_cast<uint32_t>((words + RANK_BLOCK - 1) / RANK_BLOCK);
            DeviceBuffer sums;
            HY_TRY(sums.alloc(8 * (size_t{n_blocks} + 1)));
            hipLaunchKernelGGL(rank_table_block_sums, dim3(n_blocks), dim3(256), 0, stream, entries, words, sums.as<uint64_t>());
            hipLaunchKernelGGL(scan_block_offsets, dim3(1), dim3(1024), 0, stream, sums.as<uint64_t>(), n_blocks, 0xFFFFFFFFu, sums.as<uint64_t>() + n_blocks);
            hipLaunchKernelGGL(rank_table_bases, dim3(n_blocks), dim3(256), 0, stream, entries, words, sums.as<uint64_t>());
            HY_TRY(b.rows_tmp.alloc(row_bytes * total));
            if (key32 && id32) hipLaunchKernelGGL((rank_table_scatter_rows<uint32_t, uint32_t>), key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), b.rows.as<uint32_t>(), b.rows_tmp.as<uint32_t>(), total, key_min, entries);
            else if (key32) hipLaunchKernelGGL((rank_table_scatter_rows<uint32_t, hy_row_id>), key_grid, dim3(256), 0, stream, b.keys.as<uint32_t>(), b.rows.as<hy_row_id>(), b.rows_tmp.as<hy_row_id>(), total, key_min, entries);
            else if (id32) hipLaunchKernelGGL((rank_table_scatter_rows<uint64_t, uint32_t>), key_grid, dim3(256), 0, stream, b.keys.as<uint64_t>(), b.rows.as<uint32_t>(), b.rows_tmp.as<uint32_t>(), total, key_min, entries);
            else hipLaunchKernelGGL((rank_table_scatter_rows<uint64_t, hy_row_id>), key_grid, dim3(256), 0, stream, b.keys.as<uint64_t>(), b.rows.as<hy_row_id>(), b.rows_tmp.as<hy_row_id>(), total, key_min, entries);
            std::swap(b.rows.ptr, b.rows_tmp.ptr);
            std::swap(b.rows.capacity, b.rows_tmp.capacity);
            rank_table = true;
          }
        }
        if (rank_table) {
          b.rank.entries = entries;
          b.rank.key_min = key_min;
          b.rank.range = range;
          // a dense, sorted build column with equally sized chunks: the build row of rank r is row r of the table
          bool uniform = dense && sorted_signed && build->n_chunks > 0 && build->host_segments[0].size > 0;
          for (uint32_t c = 1; c < build->n_chunks && uniform; ++c) {
            const uint32_t size = build->host_segments[c].size, first_size = build->host_segments[0].size;
            uniform = c + 1 == build->n_chunks ? size <= first_size : size == first_size;
          }
          if (uniform && FIXED_JOIN_IDENTITY) {
            b.rank.identity_rows = build->host_segments[0].size;
            b.rank.identity_inverse = 1.0 / static_cast<double>(b.rank.identity_rows);
          }
        }
      }
    }
    if (rank_table) {
      Directory& d = b.directory;   // only the RowIDs by rank
      d.keys = nullptr;
      d.keys32 = nullptr;
      d.row_ids = id32 ? nullptr : b.rows.as<hy_row_id>();
      d.ids32 = id32 ? b.rows.as<uint32_t>() : nullptr;
      d.n = total;
      d.key_min = mailbox->key_min;
      d.key_max = mailbox->key_max;
      d.shift = 0;
      d.n_buckets = 1;
      d.dir = nullptr;
      return HY_OK;
    }
    if (mailbox->unsorted) {   // not sorted: stable LSD radix sort, only over the bytes that are not constant zero
      const uint64_t key_bits_or = mailbox->key_or;
      HY_TRY(b.keys_tmp.alloc(key_bytes * (total + 4)));
      HY_TRY(b.rows_tmp.alloc(row_bytes * total));
      const bool staged = key32 && id32 && lds_atomics_are_lane_ordered(stream);   // (sort_scatter_staged: 8192-element tiles, runs instead of scattered stores)
      const uint32_t n_tiles = static_cast<uint32_t>(staged ? (total + SORT_BIG_TILE - 1) / SORT_BIG_TILE : (total + SORT_TILE - 1) / SORT_TILE);
      if (staged) {
        static OncePerDevice sort_lds_raised;
        uint64_t sort_device_bit = 0;
        if (sort_lds_raised.pending(&sort_device_bit)) {
          HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_scatter_staged), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sort_staged_lds_bytes())));
          sort_lds_raised.done(sort_device_bit);
        }
      }
      DeviceBuffer hist, bases;
      HY_TRY(hist.alloc(4 * size_t{256} * n_tiles));
      HY_TRY(bases.alloc(8 * (size_t{256} * n_tiles + 1)));
      void* src_keys = b.keys.ptr;
      void* src_rows = b.rows.ptr;
      void* dst_keys = b.keys_tmp.ptr;
      void* dst_rows = b.rows_tmp.ptr;
      for (uint32_t shift = 0; shift < (key32 ? 32u : 64u); shift += 8) {
        // a byte that is zero in every key does not move anything (64-bit keys: unless one is negative -- then the sign
        // extension is part of the order)
        if (((key_bits_or >> shift) & 0xFF) == 0 && (key32 || !(key_bits_or >> 63))) continue;
        if (staged) hipLaunchKernelGGL(sort_histogram_big, dim3(n_tiles), dim3(SORT_BIG_THREADS), 0, stream, static_cast<const uint32_t*>(src_keys), total, shift, hist.as<uint32_t>(), n_tiles);
        else if (key32) hipLaunchKernelGGL(sort_histogram<uint32_t>, dim3(n_tiles), dim3(256), 0, stream, static_cast<const uint32_t*>(src_keys), total, shift, hist.as<uint32_t>(), n_tiles);
        else hipLaunchKernelGGL(sort_histogram<uint64_t>, dim3(n_tiles), dim3(256), 0, stream, static_cast<const uint64_t*>(src_keys), total, shift, hist.as<uint32_t>(), n_tiles);
        HY_TRY(exclusive_scan(hist.as<uint32_t>(), bases.as<uint64_t>(), uint64_t{256} * n_tiles, stream));
        if (staged) hipLaunchKernelGGL(sort_scatter_staged, dim3(n_tiles), dim3(SORT_BIG_THREADS), sort_staged_lds_bytes(), stream, static_cast<const uint32_t*>(src_keys), static_cast<const uint32_t*>(src_rows), static_cast<uint32_t*>(dst_keys), static_cast<uint32_t*>(dst_rows), total, shift, bases.as<uint64_t>(), n_tiles);
        else if (key32 && id32) hipLaunchKernelGGL((sort_scatter<uint32_t, uint32_t>), dim3(n_tiles), dim3(256), 0, stream, static_cast<const uint32_t*>(src_keys), static_cast<const uint32_t*>(src_rows), static_cast<uint32_t*>(dst_keys), static_cast<uint32_t*>(dst_rows), total, shift, bases.as<uint64_t>(), n_tiles);
        else if (key32) hipLaunchKernelGGL((sort_scatter<uint32_t, hy_row_id>), dim3(n_tiles), dim3(256), 0, stream, static_cast<const uint32_t*>(src_keys), static_cast<const hy_row_id*>(src_rows), static_cast<uint32_t*>(dst_keys), static_cast<hy_row_id*>(dst_rows), total, shift, bases.as<uint64_t>(), n_tiles);
        else if (id32) hipLaunchKernelGGL((sort_scatter<uint64_t, uint32_t>), dim3(n_tiles), dim3(256), 0, stream, static_cast<const uint64_t*>(src_keys), static_cast<const uint32_t*>(src_rows), static_cast<uint64_t*>(dst_keys), static_cast<uint32_t*>(dst_rows), total, shift, bases.as<uint64_t>(), n_tiles);
        else hipLaunchKernelGGL((sort_scatter<uint64_t, hy_row_id>), dim3(n_tiles), dim3(256), 0, stream, static_cast<const uint64_t*>(src_keys), static_cast<const hy_row_id*>(src_rows), static_cast<uint64_t*>(dst_keys), static_cast<hy_row_id*>(dst_rows), total, shift, bases.as<uint64_t>(), n_tiles);
        std::swap(src_keys, dst_keys);
        std::swap(src_rows, dst_rows);
      }
      if (src_keys != b.keys.ptr) {
        std::swap(b.keys.ptr, b.keys_tmp.ptr);
        std::swap(b.keys.capacity, b.keys_tmp.capacity);
        std::swap(b.rows.ptr, b.rows_tmp.ptr);
        std::swap(b.rows.capacity, b.rows_tmp.capacity);
      }
      if (key32) HY_HIP(hipMemsetAsync(b.keys.as<uint32_t>() + total, 0, 16, stream));
    }
  }
#undef HY_JOIN_BUILD_KERNEL
  // directory
  Directory& d = b.directory;
  d.keys = key32 ? nullptr : b.keys.as<uint64_t>();
  d.keys32 = key32 ? b.keys.as<uint32_t>() : nullptr;
  d.row_ids = id32 ? nullptr : b.rows.as<hy_row_id>();
  d.ids32 = id32 ? b.rows.as<uint32_t>() : nullptr;
  d.n = total;
  d.key_min = d.key_max = 0;
  d.shift = 0;
  d.n_buckets = 1;
  if (total) {
    if (keys_were_sorted) {
      d.key_min = first_key;
      d.key_max = last_key;
    } else {   // the ends of the sorted keys
      JoinMailbox* mailbox = nullptr;
      JoinMailbox* mailbox_dev = nullptr;
      HY_TRY(join_mailbox(&mailbox, &mailbox_dev));
      if (key32) hipLaunchKernelGGL(publish_build_flags<uint32_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint32_t>(), total, mailbox_dev);
      else hipLaunchKernelGGL(publish_build_flags<uint64_t>, dim3(1), dim3(1), 0, stream, b.flags.as<uint32_t>(), b.keys.as<uint64_t>(), total, mailbox_dev);
      HY_HIP(hipStreamSynchronize(stream));
      d.key_min = mailbox->first_key;
      d.key_max = mailbox->last_key;
    }
    uint32_t buckets = 1;
    while (buckets < total && buckets < (1u << 27)) buckets <<= 1;   // 1-2 keys per bucket on uniform keys: one probe of four keys
    const uint64_t range = d.key_max - d.key_min;
    uint32_t shift = 0;
    while (shift < 64 && (range >> shift) >= buckets) ++shift;
    d.shift = shift;
    d.n_buckets = static_cast<uint32_t>((range >> shift) + 1);
  }
  HY_TRY(b.dir.alloc(4 * (size_t{d.n_buckets} + 2)));
  d.dir = b.dir.as<uint32_t>();
  if (total) {
    const dim3 grid(static_cast<uint32_t>((total + 255) / 256));
    if (key32) hipLaunchKernelGGL(directory_fill<uint32_t>, grid, dim3(256), 0, stream, d.keys32, total, d.key_min, d.shift, d.n_buckets, b.dir.as<uint32_t>());
    else hipLaunchKernelGGL(directory_fill<uint64_t>, grid, dim3(256), 0, stream, d.keys, total, d.key_min, d.shift, d.n_buckets, b.dir.as<uint32_t>());
  } else {
    HY_HIP(hipMemsetAsync(b.dir.ptr, 0, 4 * (size_t{d.n_buckets} + 2), stream));
  }
  return HY_OK;
}

// Persistent waves that take one tile at a time (rt_stream_count), STREAM_WAVES per workgroup: as many as fit the device at once
// (a multiple of the 8 XCDs), never more than there are tiles.
static uint32_t stream_grid(uint32_t n_tiles, int workgroups_per_cu) {
  uint32_t per_cu = workgroups_per_cu > 0 ? static_cast<uint32_t>(workgroups_per_cu) : 1;
  if (FIXED_JOIN_WGS_PER_CU > 0) per_cu = static_cast<uint32_t>(FIXED_JOIN_WGS_PER_CU);
  const uint32_t resident = device_cu_count() * per_cu / 8 * 8;
  const uint32_t needed = 8 * (((n_tiles + 7) / 8 + STREAM_WAVES - 1) / STREAM_WAVES);   // a wave per tile of every XCD's share
  return std::max<uint32_t>(8, std::min<uint32_t>(resident, needed));
}

static uint32_t device_cu_count() {
  int device = 0, cus = 256;
  (void)hipGetDevice(&device);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  return static_cast<uint32_t>(cus);
}

#include "join_star.hpp"

static thread_local int t_last_join_used_rank_table = 0;   // debug / tests: which lookup structure the thread's last join built
static thread_local int t_last_join_used_pkfk = 0;         // ... and whether the kernels of join_pkfk.hpp probed it
static thread_local int t_last_join_hinted_attempt = 0;
static thread_local int t_last_join_hinted = 0;            // ... 1: the build side was filled from the column's key hint, 2: that attempt was discarded and the join ran again
static thread_local bool t_join_returned_async = false;    // run_join_once returned with its kernels queued (HY_JOIN_ASYNC)
constexpr uint32_t JOIN_TRACE_TILES = 1u << 15;
static uint64_t* g_join_trace = nullptr;
static uint32_t g_join_trace_tiles = 0;

static bool same_chunk_layout(const hy_column* a, const hy_column* b) {
  if (a->n_chunks != b->n_chunks) return false;
  for (uint32_t c = 0; c < a->n_chunks; ++c) if (a->host_segments[c].size != b->host_segments[c].size) return false;
  return true;
}

static uint32_t flip_condition(uint32_t condition) {   // flip_predicate_condition (types.cpp)
  switch (condition) {
    case HY_PRED_LESS_THAN: return HY_PRED_GREATER_THAN;
    case HY_PRED_LESS_THAN_EQUALS: return HY_PRED_GREATER_THAN_EQUALS;
    case HY_PRED_GREATER_THAN: return HY_PRED_LESS_THAN;
    case HY_PRED_GREATER_THAN_EQUALS: return HY_PRED_LESS_THAN_EQUALS;
    default: return condition;
  }
}

// Probed once per process (on the device hy_init selected): 0 unknown, 1 lane-ordered, 2 not.
static std::atomic<int> g_lds_atomic_order{0};
static bool lds_atomics_are_lane_ordered(hipStream_t stream) {
  int state = g_lds_atomic_order.load(std::memory_order_acquire);
  if (state == 0) {
    state = 2;
    if (FIXED_JOIN_ORDERED_ATOMICS) {
      uint32_t* failures = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&failures), 4) == hipSuccess) {
        uint32_t host = 1;
        bool ok = hipMemsetAsync(failures, 0, 4, stream) == hipSuccess;
        for (uint32_t seed = 0; ok && seed < 4; ++seed) hipLaunchKernelGGL(lds_atomic_order_probe, dim3(512), dim3(512), 0, stream, failures, seed);
        ok = ok && hipMemcpyAsync(&host, failures, 4, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        (void)hipFree(failures);
        if (ok && host == 0) state = 1;
      }
    }
    g_lds_atomic_order.store(state, std::memory_order_release);
  }
  return state == 1;
}


// The LSD radix sort above for other operators (aggregate.hip orders the groups of a large result by it): (key, id) pairs of 32 bits each,
// ascending by key, stable; only the bytes below `key_bits` move anything.  *keys / *ids point at the sorted arrays on return (the inputs
// or the temporaries).
hy_status sort_pairs_u32(uint32_t** keys, uint32_t** ids, uint32_t* keys_tmp, uint32_t* ids_tmp, uint64_t n, uint32_t key_bits, hipStream_t stream) {
  if (n < 2) return HY_OK;
  const bool staged = lds_atomics_are_lane_ordered(stream);
  const uint32_t n_tiles = static_cast<uint32_t>(staged ? (n + SORT_BIG_TILE - 1) / SORT_BIG_TILE : (n + SORT_TILE - 1) / SORT_TILE);
  if (staged) {
    static OncePerDevice lds_raised;
    uint64_t device_bit = 0;
    if (lds_raised.pending(&device_bit)) {
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_scatter_staged), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sort_staged_lds_bytes())));
      lds_raised.done(device_bit);
    }
  }
  DeviceBuffer hist, bases;
  HY_TRY(hist.alloc(4 * size_t{256} * n_tiles));
  HY_TRY(bases.alloc(8 * (size_t{256} * n_tiles + 1)));
  uint32_t *src_keys = *keys, *src_ids = *ids, *dst_keys = keys_tmp, *dst_ids = ids_tmp;
  for (uint32_t shift = 0; shift < 32 && shift < key_bits; shift += 8) {
    if (staged) hipLaunchKernelGGL(sort_histogram_big, dim3(n_tiles), dim3(SORT_BIG_THREADS), 0, stream, src_keys, n, shift, hist.as<uint32_t>(), n_tiles);
    else hipLaunchKernelGGL(sort_histogram<uint32_t>, dim3(n_tiles), dim3(256), 0, stream, src_keys, n, shift, hist.as<uint32_t>(), n_tiles);
    HY_TRY(exclusive_scan(hist.as<uint32_t>(), bases.as<uint64_t>(), uint64_t{256} * n_tiles, stream));
    if (staged) hipLaunchKernelGGL(sort_scatter_staged, dim3(n_tiles), dim3(SORT_BIG_THREADS), sort_staged_lds_bytes(), stream, src_keys, src_ids, dst_keys, dst_ids, n, shift, bases.as<uint64_t>(), n_tiles);
    else hipLaunchKernelGGL((sort_scatter<uint32_t, uint32_t>), dim3(n_tiles), dim3(256), 0, stream, src_keys, src_ids, dst_keys, dst_ids, n, shift, bases.as<uint64_t>(), n_tiles);
    std::swap(src_keys, dst_keys);
    std::swap(src_ids, dst_ids);
  }
  *keys = src_keys;
  *ids = src_ids;
  return HY_OK;
}

// Do the probe keys lack locality as far as the host can tell?  FrameOfReference blocks of 1- or 2-byte offsets span at most 65 536 key
// values per 2048 rows (TPC-H lineitem's order keys); 4-byte offsets and unencoded values say nothing -- foreign keys of a fact table
// (SSB lineorder) are random.
static bool probe_keys_scattered(const hy_column* probe) {
  for (uint32_t c = 0; c < probe->n_chunks; ++c) {
    const hy_segment& seg = probe->host_segments[c];
    if (!(seg.encoding == HY_ENC_FRAME_OF_REFERENCE && seg.width <= 2)) return true;
  }
  return false;
}

// ... and where the layout says nothing (4-byte values / offsets), a look at the keys themselves: 1 024 samples of 64 consecutive rows, spread over
// the column; a sample is LOCAL when its keys span fewer than 2^16 key values (their table entries lie within 16 KB).  Shuffled foreign keys:
// no sample is; TPC-H's l_orderkey as a ValueSegment: every one.  The answer stays with the column (it does not change).
__global__ __launch_bounds__(64) void sample_key_spans(const SliceView* views, uint32_t n_slices, uint32_t* local_samples) {
  const uint32_t slice = static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * n_slices) / gridDim.x);
  const SliceView view = views[slice];
  if (view.row_count < 64 || (view.kind != VIEW_INT32 && view.kind != VIEW_FOR32)) return;
  const uint32_t row = view.row_begin + ((blockIdx.x * 2654435761u) % (view.row_count - 63)) + threadIdx.x;
  uint32_t key = static_cast<const uint32_t*>(view.data)[row];
  if (view.kind == VIEW_FOR32) key += static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[row / HY_FOR_BLOCK_SIZE]);
  int32_t low = static_cast<int32_t>(key), high = static_cast<int32_t>(key);
  for (int d = 32; d > 0; d >>= 1) {
    const int32_t other_low = __shfl_xor(low, d), other_high = __shfl_xor(high, d);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
  }
  if (threadIdx.x == 0 && static_cast<int64_t>(high) - static_cast<int64_t>(low) < 65536) atomicAdd(local_samples, 1u);
}

// true: the probe column's neighbouring rows do NOT hold neighbouring keys (pass 1 of an Inner join hands the ranks to pass 2)
static hy_status probe_keys_lack_locality(const hy_column* probe, hipStream_t stream, bool* scattered) {
  *scattered = false;
  if (!probe_keys_scattered(probe) || !probe->d_slice_views || probe->n_slices == 0) return HY_OK;   // (1- and 2-byte offsets: local by construction)
  uint32_t known = probe->join_hint.probe_locality.load(std::memory_order_relaxed);
  if (known == 0) {
    constexpr uint32_t SAMPLES = 1024;
    uint32_t* host = nullptr;
    uint32_t* mapped = nullptr;
    HY_TRY(pinned_staging(64, reinterpret_cast<void**>(&host), reinterpret_cast<void**>(&mapped)));
    DeviceBuffer counter;   // (the samples' atomics go to device memory: a thousand atomics on pinned host memory cross PCIe one by one, 0.75 ms)
    HY_TRY(counter.alloc(64));
    HY_HIP(hipMemsetAsync(counter.ptr, 0, 4, stream));
    hipLaunchKernelGGL(sample_key_spans, dim3(SAMPLES), dim3(64), 0, stream, probe->d_slice_views, probe->n_slices, counter.as<uint32_t>());
    HY_HIP(hipMemcpyAsync(host, counter.ptr, 4, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
    known = *host * 4 >= SAMPLES * 3 ? 1u : 2u;   // three samples in four local: clustered
    probe->join_hint.probe_locality.store(known, std::memory_order_relaxed);
  }
  *scattered = known == 2;
  return HY_OK;
}

static hy_status run_join_once(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result, bool count_only, uint64_t* count_out,
                               const hy_join_predicate* secondary, uint32_t n_secondary, bool* retry) {
  if (mode == HY_JOIN_FULL_OUTER || mode == HY_JOIN_CROSS || mode > HY_JOIN_ANTI_NULL_AS_FALSE) return fail(HY_ERR_UNSUPPORTED, "JoinHash does not support join mode %u (join_hash.cpp:38-44)", mode);
  if (n_secondary > HY_MAX_SECONDARY_PREDICATES) return fail(HY_ERR_UNSUPPORTED, "more than %u secondary join predicates stay on the CPU path", HY_MAX_SECONDARY_PREDICATES);
  if (n_secondary && mode == HY_JOIN_ANTI_NULL_AS_TRUE) return fail(HY_ERR_UNSUPPORTED, "JoinHash does not support secondary predicates with AntiNullAsTrue (join_hash.cpp:39-44)");
  for (uint32_t p = 0; p < n_secondary; ++p) {
    const hy_join_predicate& predicate = secondary[p];
    if (!predicate.left_column || !predicate.right_column) return fail(HY_ERR_INVALID, "secondary join predicate %u: column missing", p);
    if (predicate.condition > HY_PRED_GREATER_THAN_EQUALS) return fail(HY_ERR_INVALID, "secondary join predicate %u: condition %u is no comparison", p, predicate.condition);
    for (const hy_column* column : {predicate.left_column, predicate.right_column}) {
      if (column->is_mvcc || (column->ref && column->ref->is_mvcc)) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
      if (column->data_type < HY_TYPE_INT || column->data_type > HY_TYPE_DOUBLE) return fail(HY_ERR_UNSUPPORTED, "secondary join predicates on strings stay on the CPU path");
    }
    if (!same_chunk_layout(predicate.left_column, left) || !same_chunk_layout(predicate.right_column, right))
      return fail(HY_ERR_INVALID, "secondary join predicate %u: the columns do not have the chunk layout of the join's input tables", p);
  }
  for (const hy_column* column : {left, right}) {
    if (column->is_mvcc || (column->ref && column->ref->is_mvcc)) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
    if (column->data_type < HY_TYPE_INT || column->data_type > HY_TYPE_DOUBLE) return fail(HY_ERR_UNSUPPORTED, "string join keys stay on the CPU path");
  }
  // JoinHashTraits (join_hash_traits.hpp:15-40): the type both sides are cast to; 0 = an integer type (std::hash is the identity)
  uint32_t hashed_type = 0;
  {
    const uint32_t l = left->data_type, r = right->data_type;
    const bool l_float = l == HY_TYPE_FLOAT || l == HY_TYPE_DOUBLE, r_float = r == HY_TYPE_FLOAT || r == HY_TYPE_DOUBLE;
    if (l_float && r_float) hashed_type = (l == HY_TYPE_DOUBLE || r == HY_TYPE_DOUBLE) ? HY_TYPE_DOUBLE : HY_TYPE_FLOAT;
    else if (l_float || r_float) hashed_type = l_float ? l : r;
  }
  const bool general = n_secondary != 0 || hashed_type != 0;   // the <true> instantiations of the probe kernels
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
  // RowID -> chunk_id << 16 | chunk_offset needs both below 2^16 (every table with Hyrise's default chunk size)
  bool pack_build_ids = !semi_anti && !count_only && build->n_chunks <= 65536;
  for (uint32_t c = 0; c < build->n_chunks && pack_build_ids; ++c) pack_build_ids = build->host_segments[c].size <= 65536;
  BuildSide b;
  StageClock clock;
  // Will the probe side take the kernels of join_pkfk.hpp if the build side turns out to be a rank table?  (Everything that does not
  // depend on the build side: probe segments that SliceViews describe -- int32 values / FrameOfReference offsets, no NULLs, 16-byte
  // aligned --, counts and pair indices in 32 bits, lane-ordered LDS atomics.)
  bool probe_views = !probe->is_reference && probe->n_slices > 0 && FIXED_JOIN_FETCH_AHEAD;
  for (uint32_t c = 0; c < probe->n_chunks && probe_views; ++c) {
    const hy_segment& seg = probe->host_segments[c];
    probe_views = !seg.nulls && reinterpret_cast<uintptr_t>(seg.data) % 16 == 0 &&
                  ((seg.encoding == HY_ENC_UNENCODED && seg.data_type == HY_TYPE_INT) || seg.encoding == HY_ENC_FRAME_OF_REFERENCE);
  }
  const bool probe_takes_pk = probe_views && hashed_type == 0 && n_secondary == 0 && probe->rows < 0xFFFF0000ull && option(HY_OPT_JOIN_PKFK) &&
                              (count_only || result->capacity <= 0xFFFFFFFFull) && lds_atomics_are_lane_ordered(stream);
  HY_TRY(prepare_build(build, keep_nulls_build, probe_filtered, pack_build_ids, hashed_type, n_secondary == 0, semi_anti && n_secondary == 0, probe_takes_pk, b, stream));
  const bool rank_path = b.rank.entries != nullptr;
  // A rank table filled from the build column's key hint (rank_table_fill_checked) is confirmed when the join's kernels have finished:
  // if the column is not what the hint said, the hint is dropped, whatever the join wrote is discarded and the caller runs it again.
  auto hint_confirmed = [&](const JoinMailbox* mailbox) {   // (after the stream synchronise that makes pk_plan's mailbox visible)
    if (!b.hinted || !mailbox->build_unconfirmed) return true;
    build->join_hint.state.store(2, std::memory_order_release);
    *retry = true;
    return false;
  };
  const bool fetch_ahead = rank_path && probe_views;   // pass 1 of the general rank-table kernels is the wave-per-tile kernel with wide loads
  t_last_join_used_rank_table = rank_path ? (b.rank.identity_rows ? 2 : 1) : 0;
  t_last_join_used_pkfk = 0;
  t_last_join_hinted_attempt = b.hinted ? 1 : 0;
  clock.mark("build side launched");

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

  // The primary-key / foreign-key probe (join_pkfk.hpp): rank table, probe segments that SliceViews describe, keys of both sides
  // int32 values (32-bit distances), counts and pair indices in 32 bits, lane-ordered LDS atomics.
  const bool pk_path = rank_path && probe_takes_pk && static_cast<int64_t>(b.rank.key_min) >= INT32_MIN && static_cast<int64_t>(b.rank.key_min + b.rank.range) <= INT32_MAX;
  if (b.bloom_is_bits && !pk_path) return fail(HY_ERR_DEVICE, "join: a bit filter was built for kernels that do not run");   // (cannot happen: a hinted build has int32 keys)
  if (pk_path) {
    const uint32_t n_tiles = probe->n_slices * PK_TILES_PER_SLICE, partitions = 1u << radix_bits;
    const uint32_t stride = (n_tiles + 1 + 3) & ~3u;
    const uint32_t n_groups = radix_bits ? partitions : probe->n_chunks;
    const uint32_t max_slices = static_cast<uint32_t>(probe->rows / PROBE_SIZE_PER_CHUNK) + n_groups + 1;
    DeviceBuffer cells, small;
    HY_TRY(cells.alloc(size_t{12} * partitions * stride));
    const size_t at_origin = align_up(8 * size_t{partitions}, 8), at_slice_base = at_origin + 8 * (size_t{partitions} + 1), at_words = align_up(at_slice_base + 4 * (size_t{n_groups} + 1), 8);
    HY_TRY(small.alloc(at_words + 64));
    JoinMailbox* mailbox = nullptr;
    JoinMailbox* mailbox_dev = nullptr;
    HY_TRY(join_mailbox(&mailbox, &mailbox_dev));
    hy_row_id* user_build = nullptr;
    hy_row_id* user_probe = nullptr;
    if (!count_only) {
      user_build = result->left_is_build ? result->left_pos : result->right_pos;
      user_probe = result->left_is_build ? result->right_pos : result->left_pos;
      if (!user_probe && result->capacity) return fail(HY_ERR_INVALID, "join result: PosList buffer for the probe side missing");
      if (!semi_anti && !user_build && result->capacity) return fail(HY_ERR_INVALID, "join result: PosList buffer for the build side missing");
      if (!result->slice_offsets) return fail(HY_ERR_INVALID, "join result: slice_offsets missing");
    }
    DeviceBuffer d_build_out, d_probe_out, d_slice_offsets;
    uint64_t* dev_slice_offsets = count_only ? nullptr : result->slice_offsets;
    if (!count_only && host_result) {
      HY_TRY(d_slice_offsets.alloc(8 * (size_t{max_slices} + 2)));
      dev_slice_offsets = d_slice_offsets.as<uint64_t>();
    }
    PkArgs k{};
    k.views = probe->d_slice_views;
    k.n_tiles = n_tiles;
    k.stride = stride;
    k.mode = mode;
    k.radix_bits = radix_bits;
    k.keep_nulls = keep_nulls_probe;
    k.n_groups = n_groups;
    k.build_bloom = probe_filtered ? b.bloom.as<uint8_t>() : nullptr;
    k.bloom_is_bits = b.bloom_is_bits ? 1u : 0u;
    k.rank = b.rank;
    k.ids32 = b.directory.ids32;
    k.row_ids = b.directory.row_ids;
    k.counts = cells.as<uint32_t>();
    k.rel_elements = k.counts + size_t{partitions} * stride;
    k.rel_pairs = k.rel_elements + size_t{partitions} * stride;
    char* small_base = small.as<char>();
    k.totals = reinterpret_cast<uint32_t*>(small_base);
    k.origin_pairs = reinterpret_cast<uint64_t*>(small_base + at_origin);
    k.slice_base = reinterpret_cast<uint32_t*>(small_base + at_slice_base);
    k.group_first_tile = probe->d_first_slice;
    k.ticket = reinterpret_cast<uint32_t*>(small_base + at_words);
    k.plan = reinterpret_cast<JoinPlan*>(small_base + at_words + 16);
    k.mailbox = mailbox_dev;
    k.capacity = count_only ? ~0ull : result->capacity;
    k.slice_capacity = count_only ? 0xFFFFFFFFu : result->slice_capacity;
    if (HY_DEBUG_ENV("HY_JOIN_TRACE")) {
      static uint64_t* trace_buffer = nullptr;
      if (!trace_buffer) (void)hipMalloc(reinterpret_cast<void**>(&trace_buffer), 8 * 6 * JOIN_TRACE_TILES);
      if (n_tiles <= JOIN_TRACE_TILES) { k.trace = trace_buffer; g_join_trace = trace_buffer; g_join_trace_tiles = n_tiles; }
    }
    k.slice_offsets = dev_slice_offsets;
    // HY_JOIN_ASYNC: no host round trip at all -- pk_plan leaves in device memory what the host would read from the mailbox
    const bool async = !count_only && !host_result && (result->flags & HY_JOIN_ASYNC) && result->status;
    k.status = async ? result->status : nullptr;
    k.verdict = b.hinted ? b.verdict : nullptr;
    k.fill_records = b.hinted ? b.fill_records : nullptr;
    k.n_fill_records = b.hinted ? b.n_fill_records : 0;
    k.hint_min = b.hint_min;
    k.hint_max = b.hint_max;
    k.hint_allows_duplicates = b.hint_allows_duplicates ? 1u : 0u;
    const uint32_t tile_grid = 8 * ((n_tiles + 7) / 8);
    // The build side staged in LDS (pk_count_lds, join_pkfk.hpp): a rank table over fewer than 2^20 key values, enough tiles for
    // persistent workgroups to pay for staging it once per CU.
    DeviceBuffer row_masks, row_ranks;
    bool hand_over_ranks = false;
    const bool build_in_lds = b.rank.range < PK_LDS_KEYS && partitions <= PK_LDS_MAX_PARTITIONS && n_tiles >= static_cast<uint64_t>(option(HY_OPT_JOIN_LDS_BUILD_TILES)) && option(HY_OPT_JOIN_LDS_BUILD);   // (tests lower the bar)
    if (build_in_lds) {
      HY_TRY(row_masks.alloc(size_t{n_tiles} * (2 * PK_TILE / 8)));
      k.row_masks = row_masks.as<uint8_t>();
      static OncePerDevice count_lds_raised;
      uint64_t device_bit = 0;
      if (count_lds_raised.pending(&device_bit)) {
        HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk_count_lds), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pk_count_lds_bytes(PK_LDS_KEYS - 1, PK_LDS_MAX_PARTITIONS))));
        count_lds_raised.done(device_bit);
      }
      int device = 0, cus = 256;
      (void)hipGetDevice(&device);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      const uint32_t groups = std::min<uint32_t>(static_cast<uint32_t>(cus), (n_tiles + PK_LDS_SUBTILES - 1) / PK_LDS_SUBTILES);
      hipEvent_t count_started = nullptr, count_stopped = nullptr;
      profile_events(&count_started, &count_stopped, HY_KERNEL_JOIN_COUNT);
      hipExtLaunchKernelGGL(pk_count_lds, dim3(groups), dim3(1024), pk_count_lds_bytes(b.rank.range, partitions), stream, count_started, count_stopped, 0, k);
    } else {
      hipEvent_t count_started = nullptr, count_stopped = nullptr;
      profile_events(&count_started, &count_stopped, HY_KERNEL_JOIN_COUNT);
      // Probe keys without locality, an Inner join over a million rows or more: pass 1 hands every row's partner rank to pass 2 (pk_count_wave RANKS)
      // (... and a table of a megabyte or more: random lookups in a smaller one stay in the L2)
      if (mode == HY_JOIN_INNER && !count_only && option(HY_OPT_JOIN_HAND_OVER_RANKS) > 0 && probe->rows >= static_cast<uint64_t>(option(HY_OPT_JOIN_HAND_OVER_RANKS)) &&
          (b.rank.range >> 5) * 8 >= (option(HY_OPT_JOIN_HAND_OVER_RANKS) == 1 ? 0u : (1u << 20)))
        HY_TRY(probe_keys_lack_locality(probe, stream, &hand_over_ranks));
      if (hand_over_ranks) {
        HY_TRY(row_ranks.alloc(4 * size_t{n_tiles} * PK_TILE));
        k.row_ranks = row_ranks.as<uint32_t>();
        hipExtLaunchKernelGGL(pk_count<true>, dim3(tile_grid), dim3(PK_COUNT_THREADS), 0, stream, count_started, count_stopped, 0, k);
      } else hipExtLaunchKernelGGL(pk_count<false>, dim3(tile_grid), dim3(PK_COUNT_THREADS), 0, stream, count_started, count_stopped, 0, k);
    }
    hipLaunchKernelGGL(pk_scan, dim3(partitions), dim3(PK_SCAN_THREADS), 0, stream, k);
    clock.mark("pass 1 launched");
    if (count_only) {
      HY_HIP(hipStreamSynchronize(stream));
      if (!hint_confirmed(mailbox)) return HY_OK;
      if (count_out) *count_out = mailbox->n_pairs;
      return HY_OK;
    }
    hy_row_id* dev_build = user_build;
    hy_row_id* dev_probe = user_probe;
    if (host_result) {   // the staging buffers are sized by the pair count: ask now
      HY_HIP(hipStreamSynchronize(stream));
      clock.mark("pass 1 done");
      if (mailbox->fits) {
        if (!semi_anti) { HY_TRY(d_build_out.alloc(8 * mailbox->n_pairs)); dev_build = d_build_out.as<hy_row_id>(); }
        HY_TRY(d_probe_out.alloc(8 * mailbox->n_pairs));
        dev_probe = d_probe_out.as<hy_row_id>();
      }
    }
    k.build_out = semi_anti ? nullptr : dev_build;
    k.probe_out = dev_probe;
    static OncePerDevice pk_lds_raised;
    uint64_t pk_device_bit = 0;
    if (pk_lds_raised.pending(&pk_device_bit)) {
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk_emit<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pk_emit_lds_words(MAX_PARTITIONS)));
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk_emit<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pk_emit_lds_words(MAX_PARTITIONS)));
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk_emit<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pk_emit_lds_words(MAX_PARTITIONS)));
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk_emit<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pk_emit_lds_words(MAX_PARTITIONS)));
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk_emit<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * pk_emit_lds_words(MAX_PARTITIONS)));
      pk_lds_raised.done(pk_device_bit);
    }
    hipEvent_t started = nullptr, stopped = nullptr;   // (stamped from the dispatch packet itself)
    profile_events(&started, &stopped, HY_KERNEL_JOIN_PROBE);
    const uint32_t cut_grid = std::min<uint32_t>(max_slices, result->slice_capacity);
    {
      const int64_t group = FIXED_JOIN_EMIT_TILE_GROUP;
      k.emit_group_shift = 32;
      if (group > 0) { k.emit_group_shift = 0; while (k.emit_group_shift < 16 && (int64_t{1} << (k.emit_group_shift + 1)) <= group) ++k.emit_group_shift; }
    }
    k.cut_blocks = (cut_grid + 7) / 8 * 8;   // (pk_cut_slice returns at once for slices the plan does not have)
    k.cleaning = b.cleaning;
    if (build_in_lds) {   // (pass 2 reads the rows' found / materialised bits pass 1 left behind)
      if (mode == HY_JOIN_INNER) hipExtLaunchKernelGGL((pk_emit<true, true>), dim3(k.cut_blocks + tile_grid), dim3(PK_THREADS), 4 * pk_emit_lds_words(partitions), stream, started, stopped, 0, k);
      else hipExtLaunchKernelGGL((pk_emit<false, true>), dim3(k.cut_blocks + tile_grid), dim3(PK_THREADS), 4 * pk_emit_lds_words(partitions), stream, started, stopped, 0, k);
    } else if (hand_over_ranks) hipExtLaunchKernelGGL((pk_emit<true, false, true>), dim3(k.cut_blocks + tile_grid), dim3(PK_THREADS), 4 * pk_emit_lds_words(partitions), stream, started, stopped, 0, k);
    else if (mode == HY_JOIN_INNER) hipExtLaunchKernelGGL(pk_emit<true>, dim3(k.cut_blocks + tile_grid), dim3(PK_THREADS), 4 * pk_emit_lds_words(partitions), stream, started, stopped, 0, k);
    else hipExtLaunchKernelGGL(pk_emit<false>, dim3(k.cut_blocks + tile_grid), dim3(PK_THREADS), 4 * pk_emit_lds_words(partitions), stream, started, stopped, 0, k);
    HY_HIP(hipGetLastError());
    if (b.cleaned) { b.cleaned->clean = true; b.cleaned->dirty_table = b.cleaned->dirty_filter = 0; }
    clock.mark("pass 2 launched");
    t_last_join_used_pkfk = build_in_lds ? 2 : 1;   // debug / tests: the primary-key / foreign-key kernels ran (2: with the build side's bits staged in LDS)
    if (host_result) {
      if (mailbox->fits && mailbox->n_pairs) {
        HY_HIP(hipMemcpyAsync(user_probe, dev_probe, 8 * mailbox->n_pairs, hipMemcpyDeviceToHost, stream));
        if (!semi_anti) HY_HIP(hipMemcpyAsync(user_build, dev_build, 8 * mailbox->n_pairs, hipMemcpyDeviceToHost, stream));
      }
      if (mailbox->fits) HY_HIP(hipMemcpyAsync(result->slice_offsets, dev_slice_offsets, 8 * (size_t{mailbox->n_slices} + 1), hipMemcpyDeviceToHost, stream));
    }
    if (async) {   // the temporaries go back to this thread's pool: whatever takes them next is queued behind these kernels on the same stream
      t_join_returned_async = true;
      return HY_OK;
    }
    HY_HIP(hipStreamSynchronize(stream));   // the temporaries above are freed on return
    clock.mark("pass 2 done");
    if (!hint_confirmed(mailbox)) return HY_OK;
    result->n_slices = mailbox->n_slices;
    result->n_pairs = mailbox->n_pairs;
    if (mailbox->n_slices > result->slice_capacity) return fail(HY_ERR_CAPACITY, "join produces %u output PosLists, slice capacity is %u", mailbox->n_slices, result->slice_capacity);
    if (mailbox->n_pairs > result->capacity) return fail(HY_ERR_CAPACITY, "join produces %llu pairs, capacity is %llu", static_cast<unsigned long long>(mailbox->n_pairs), static_cast<unsigned long long>(result->capacity));
    return HY_OK;
  }

  // probe pass 1
  const uint32_t n_tiles = probe->n_slices * (SLICE_ROWS / JOIN_TILE);
  const uint32_t partitions = 1u << radix_bits;
  const size_t cells = size_t{partitions} * n_tiles;
  // pass 1's per-(partition, tile) counts: elements | zeros up to a scan-block boundary | pairs -- one buffer, one scan
  const uint64_t second_at = (uint64_t{cells} + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK * SCAN_BLOCK;
  DeviceBuffer hist, base;
  HY_TRY(hist.alloc(4 * (second_at + cells + 1)));
  HY_TRY(base.alloc(8 * (second_at + cells + 2)));
  ProbeArgs a{};
  a.segments = probe->d_segments;
  a.slices = probe->d_slices;
  a.views = probe->d_slice_views;
  a.n_tiles = n_tiles;
  a.mode = mode;
  a.radix_bits = radix_bits;
  a.keep_nulls = keep_nulls_probe;
  a.build_rows_zero = build->rows == 0;
  a.build_bloom = probe_filtered ? b.bloom.as<uint8_t>() : nullptr;
  a.dir = b.directory;
  a.rank = b.rank;
  a.trace = nullptr;
  if (HY_DEBUG_ENV("HY_JOIN_TRACE")) {
    static uint64_t* trace_buffer = nullptr;
    if (!trace_buffer) (void)hipMalloc(reinterpret_cast<void**>(&trace_buffer), 8 * 6 * JOIN_TRACE_TILES);
    if (n_tiles <= JOIN_TRACE_TILES) { a.trace = trace_buffer; g_join_trace = trace_buffer; g_join_trace_tiles = n_tiles; }
  }
  a.pack_build_ids = b.directory.ids32 ? 1 : 0;
  a.hashed_type = hashed_type;
  a.n_secondary = n_secondary;
  for (uint32_t p = 0; p < n_secondary; ++p) {   // as the probe sees them: build <condition> probe (join_hash.cpp:158-165)
    a.secondary[p].build = (build_right ? secondary[p].right_column : secondary[p].left_column)->d_segments;
    a.secondary[p].probe = (build_right ? secondary[p].left_column : secondary[p].right_column)->d_segments;
    a.secondary[p].condition = build_right ? flip_condition(secondary[p].condition) : secondary[p].condition;
  }
  a.hist_elements = hist.as<uint32_t>();
  a.hist_pairs = hist.as<uint32_t>() + second_at;
  a.base_elements = base.as<uint64_t>();
  a.base_pairs = base.as<uint64_t>() + second_at;
  DeviceBuffer d_partner, d_meta, d_uncached;
  if (!count_only && n_tiles && !rank_path) {   // pass 2 follows: let pass 1 leave its lookup results behind (6 B per probe row; a rank table is looked up again)
    HY_TRY(d_partner.alloc(4 * size_t{n_tiles} * JOIN_TILE));
    HY_TRY(d_meta.alloc(2 * size_t{n_tiles} * JOIN_TILE));
    HY_TRY(d_uncached.alloc(4 * size_t{n_tiles} * 2));   // flags | work list of probe_emit_generic
    a.uncached_tiles = d_uncached.as<uint32_t>() + n_tiles;
    a.row_partner = d_partner.as<uint32_t>();
    a.row_meta = d_meta.as<uint32_t>();
    a.tile_uncached = d_uncached.as<uint32_t>();
  }
  JoinMailbox* mailbox = nullptr;
  JoinMailbox* mailbox_dev = nullptr;
  HY_TRY(join_mailbox(&mailbox, &mailbox_dev));
  DeviceBuffer d_words;   // [0] pass 1's error flag (unused), [1] tiles with multi-partner rows, [8..15] XCD tickets, [16..17] JoinPlan
  HY_TRY(d_words.alloc(256));
  hipLaunchKernelGGL(zero_two, dim3(static_cast<uint32_t>(std::min<uint64_t>(64, (second_at - cells + 32 + 255) / 256))), dim3(256), 0, stream, hist.as<uint32_t>() + cells, size_t{second_at - cells},
                     d_words.as<uint32_t>(), size_t{32});   // the zeros between pass 1's two count arrays | the words above
  a.error = &mailbox_dev->error;
  a.n_uncached = d_words.as<uint32_t>() + 1;
  a.xcd_tickets = d_words.as<uint32_t>() + 8;
  JoinPlan* d_plan = reinterpret_cast<JoinPlan*>(d_words.as<uint32_t>() + 16);
  a.plan = d_plan;

  // Groups of the output: the radix partitions, or -- no radix partitioning -- the probe chunks (their tiles are
  // consecutive).  Every 131 070 materialised elements of a group start a new output PosList.
  const uint32_t n_groups = radix_bits ? partitions : probe->n_chunks;
  DeviceBuffer d_first_cell, d_origin, d_slice_base;
  HY_TRY(d_origin.alloc(8 * (size_t{n_groups} + 1)));
  HY_TRY(d_slice_base.alloc(4 * (size_t{n_groups} + 1)));
  std::vector<uint64_t> group_first_cell;   // (source of an asynchronous upload: lives until the join returns)
  const uint64_t* dev_first_cell = nullptr;   // radix partitions: group g starts at cell g * n_tiles
  if (!radix_bits) {
    group_first_cell.assign(size_t{n_groups} + 1, 0);
    uint64_t tile = 0;
    for (uint32_t c = 0; c < probe->n_chunks; ++c) {
      const uint32_t chunk_slices = (probe->host_segments[c].size + SLICE_ROWS - 1) / SLICE_ROWS;
      group_first_cell[c] = tile;
      tile += (chunk_slices ? chunk_slices : 1) * (SLICE_ROWS / JOIN_TILE);   // the tiles of each of its slices, consecutive
    }
    group_first_cell[n_groups] = tile;
    HY_TRY(d_first_cell.alloc(8 * (size_t{n_groups} + 1)));
    HY_HIP(hipMemcpyAsync(d_first_cell.ptr, group_first_cell.data(), 8 * (size_t{n_groups} + 1), hipMemcpyHostToDevice, stream));
    dev_first_cell = d_first_cell.as<uint64_t>();
  }
  const uint32_t max_slices = static_cast<uint32_t>(probe->rows / PROBE_SIZE_PER_CHUNK) + n_groups + 1;   // what the host knows without asking

  // the caller's buffers
  hy_row_id* user_build = nullptr;
  hy_row_id* user_probe = nullptr;
  if (!count_only) {
    user_build = result->left_is_build ? result->left_pos : result->right_pos;
    user_probe = result->left_is_build ? result->right_pos : result->left_pos;
    if (!user_probe && result->capacity) return fail(HY_ERR_INVALID, "join result: PosList buffer for the probe side missing");
    if (!semi_anti && !user_build && result->capacity) return fail(HY_ERR_INVALID, "join result: PosList buffer for the build side missing");
    if (!result->slice_offsets) return fail(HY_ERR_INVALID, "join result: slice_offsets missing");
  }
  DeviceBuffer d_build_out, d_probe_out, d_slice_offsets;
  uint64_t* dev_slice_offsets = count_only ? nullptr : result->slice_offsets;
  if (!count_only && host_result) {
    HY_TRY(d_slice_offsets.alloc(8 * (size_t{max_slices} + 2)));
    dev_slice_offsets = d_slice_offsets.as<uint64_t>();
  }

  // pass 1, the scan of its counts, the plan of the output -- no host round trip in between
  if (n_tiles) {
    if (rank_path && fetch_ahead) {
      int per_cu = 0;
      HY_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(rt_stream_count), 64 * STREAM_WAVES, 4 * stream_count_lds_words(partitions)));
      hipLaunchKernelGGL(rt_stream_count, dim3(stream_grid(n_tiles, per_cu)), dim3(64 * STREAM_WAVES), 4 * stream_count_lds_words(partitions), stream, a);
    } else if (rank_path) {
      hipLaunchKernelGGL(rt_probe_count, dim3(probe_grid(n_tiles)), dim3(JOIN_THREADS), 0, stream, a);
    } else if (general) hipLaunchKernelGGL(probe_count<true>, dim3(probe_grid(n_tiles)), dim3(JOIN_THREADS), 0, stream, a);
    else hipLaunchKernelGGL(probe_count<false>, dim3(probe_grid(n_tiles)), dim3(JOIN_THREADS), 0, stream, a);
    HY_TRY(exclusive_scan(hist.as<uint32_t>(), base.as<uint64_t>(), second_at + cells, stream, second_at));
  } else {
    HY_HIP(hipMemsetAsync(base.ptr, 0, 8 * (second_at + cells + 2), stream));
  }
  hipLaunchKernelGGL(plan_output, dim3(1), dim3(256), 0, stream, a.base_elements, a.base_pairs, dev_first_cell, n_groups, n_tiles, uint64_t{cells},
                     count_only ? ~0ull : result->capacity, count_only ? 0xFFFFFFFFu : result->slice_capacity, d_origin.as<uint64_t>(), d_slice_base.as<uint32_t>(),
                     dev_slice_offsets, n_tiles && a.row_meta ? a.n_uncached : nullptr, d_plan, mailbox_dev);
  clock.mark("pass 1 launched");
  if (count_only) {
    HY_HIP(hipStreamSynchronize(stream));
    if (mailbox->error) return fail(HY_ERR_UNSUPPORTED, "a probe row matches more than 4 194 303 build rows");   // (pass 1 clamped its count)
    if (count_out) *count_out = mailbox->n_pairs;
    return HY_OK;
  }
  hy_row_id* dev_build = user_build;
  hy_row_id* dev_probe = user_probe;
  if (host_result) {   // the staging buffers are sized by the pair count: ask now
    HY_HIP(hipStreamSynchronize(stream));
    clock.mark("pass 1 done");
    if (mailbox->fits) {
      if (!semi_anti) { HY_TRY(d_build_out.alloc(8 * mailbox->n_pairs)); dev_build = d_build_out.as<hy_row_id>(); }
      HY_TRY(d_probe_out.alloc(8 * mailbox->n_pairs));
      dev_probe = d_probe_out.as<hy_row_id>();
    }
  }
  a.partition_element_origin = d_origin.as<uint64_t>();
  a.partition_slice_base = d_slice_base.as<uint32_t>();
  a.build_out = semi_anti ? nullptr : dev_build;
  a.probe_out = dev_probe;
  a.slice_offsets = dev_slice_offsets;
  static OncePerDevice lds_raised;   // (joins run from any thread; setting the attributes twice is harmless)
  uint64_t lds_device_bit = 0;
  if (n_tiles && lds_raised.pending(&lds_device_bit)) {
    HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_emit_cached), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * probe_emit_cached_lds_words(MAX_PARTITIONS)));
    HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_emit_generic<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * probe_emit_lds_words(MAX_PARTITIONS)));
    HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_emit_generic<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * probe_emit_lds_words(MAX_PARTITIONS)));
    HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rt_probe_emit), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * rt_probe_emit_lds_words(MAX_PARTITIONS)));
    lds_raised.done(lds_device_bit);
  }
  if (n_tiles && rank_path) {   // (every pass 2 kernel returns at once if the plan says that the result does not fit)
    a.lane_ordered_atomics = lds_atomics_are_lane_ordered(stream) ? 1u : 0u;
    hipEvent_t started = nullptr, stopped = nullptr;   // (stamped from the dispatch packet itself: event records around the launch add the gaps to its neighbours)
    profile_events(&started, &stopped, HY_KERNEL_JOIN_PROBE);
    hipExtLaunchKernelGGL(rt_probe_emit, dim3(probe_grid(n_tiles)), dim3(JOIN_THREADS), 4 * rt_probe_emit_lds_words(partitions), stream, started, stopped, 0, a);
    const uint32_t cut_grid = std::min<uint32_t>(max_slices, result->slice_capacity);
    if (cut_grid) hipLaunchKernelGGL(rt_probe_cuts, dim3(cut_grid), dim3(JOIN_THREADS), 0, stream, a, dev_first_cell, n_groups);
  } else if (n_tiles) {
    uint32_t workgroups_per_cu = 1;
    {
      int per_cu = 0;
      HY_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(probe_emit_cached), JOIN_THREADS, 4 * probe_emit_cached_lds_words(partitions)));
      workgroups_per_cu = per_cu > 0 ? static_cast<uint32_t>(per_cu) : 1;
    }
    if (FIXED_JOIN_WGS_PER_CU > 0) workgroups_per_cu = static_cast<uint32_t>(FIXED_JOIN_WGS_PER_CU);
    // persistent workgroups: as many as fit the device at once (a multiple of the 8 XCDs), never more than tiles
    const uint32_t resident = device_cu_count() * workgroups_per_cu / 8 * 8;
    const uint32_t cached_grid = std::max<uint32_t>(8, std::min<uint32_t>(resident, probe_grid(n_tiles)));
    profile_begin(stream, HY_KERNEL_JOIN_PROBE);
    hipLaunchKernelGGL(probe_emit_cached, dim3(cached_grid), dim3(JOIN_THREADS), 4 * probe_emit_cached_lds_words(partitions), stream, a);
    profile_end(stream);
    const uint32_t cut_grid = std::min<uint32_t>(max_slices, result->slice_capacity);
    if (cut_grid) hipLaunchKernelGGL(probe_cuts, dim3(cut_grid), dim3(64), 0, stream, a, dev_first_cell, n_groups);
    // the tiles pass 1 listed (usually none: the workgroups read the list's length and leave)
    if (!host_result || mailbox->n_uncached) {
      const dim3 generic_grid(std::min<uint32_t>(n_tiles, device_cu_count() * 2));
      if (general) hipLaunchKernelGGL(probe_emit_generic<true>, generic_grid, dim3(JOIN_THREADS), 4 * probe_emit_lds_words(partitions), stream, a);
      else hipLaunchKernelGGL(probe_emit_generic<false>, generic_grid, dim3(JOIN_THREADS), 4 * probe_emit_lds_words(partitions), stream, a);
    }
  }
  HY_HIP(hipGetLastError());
  clock.mark("pass 2 launched");
  if (host_result) {
    if (mailbox->fits && mailbox->n_pairs) {
      HY_HIP(hipMemcpyAsync(user_probe, dev_probe, 8 * mailbox->n_pairs, hipMemcpyDeviceToHost, stream));
      if (!semi_anti) HY_HIP(hipMemcpyAsync(user_build, dev_build, 8 * mailbox->n_pairs, hipMemcpyDeviceToHost, stream));
    }
    if (mailbox->fits) HY_HIP(hipMemcpyAsync(result->slice_offsets, dev_slice_offsets, 8 * (size_t{mailbox->n_slices} + 1), hipMemcpyDeviceToHost, stream));
  }
  HY_HIP(hipStreamSynchronize(stream));   // the temporaries above are freed on return
  clock.mark("pass 2 done");
  result->n_slices = mailbox->n_slices;
  result->n_pairs = mailbox->n_pairs;
  if (mailbox->n_slices > result->slice_capacity) return fail(HY_ERR_CAPACITY, "join produces %u output PosLists, slice capacity is %u", mailbox->n_slices, result->slice_capacity);
  if (mailbox->n_pairs > result->capacity) return fail(HY_ERR_CAPACITY, "join produces %llu pairs, capacity is %llu", static_cast<unsigned long long>(mailbox->n_pairs), static_cast<unsigned long long>(result->capacity));
  if (mailbox->error) return fail(HY_ERR_UNSUPPORTED, "a probe row matches more than 4 194 303 build rows");
  return HY_OK;
}

// JoinHash materialises its inputs (join_hash_steps.hpp:274-330): a reference input -- the output of a scan or of an earlier join, 8 bytes of
// RowID in front of every key -- is read through its PosLists ONCE, into a plain int32 column with the input's chunk layout (every chunk on
// a 16-byte boundary), and the join runs on that: its probe kernels stream value columns (the primary-key / foreign-key path takes them),
// where a reference column goes row by row through the generic decoders (orders x a scan's 25.8 M lineitems: 1.39 -> 0.5 ms).  The pairs are
// positions in the input tables either way.  Not for inputs that hold a NULL (checked on the device: the twin carries no null vector).
__global__ __launch_bounds__(256) void join_any_null_byte(const uint8_t* bytes, uint64_t n, uint32_t* found) {
  bool any = false;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) any = any || bytes[i] != 0;
  if (__any(any) && (threadIdx.x & 63) == 0) *found = 1;
}
struct MaterialisedInput {
  DeviceBuffer values, nulls, bases;
  hy_column* column = nullptr;
  MaterialisedInput() = default;
  MaterialisedInput(const MaterialisedInput&) = delete;
  MaterialisedInput& operator=(const MaterialisedInput&) = delete;
  ~MaterialisedInput() { if (column) (void)hy_column_destroy(column); }
};
constexpr uint64_t MATERIALISE_FROM_ROWS = 1u << 17;   // (smaller inputs: the extra launches cost what they save)
static hy_status materialise_reference_input(const hy_column* input, MaterialisedInput& out) {
  if (!input->is_reference || input->data_type != HY_TYPE_INT || input->rows < MATERIALISE_FROM_ROWS || input->has_dictionary_without_values || input->is_mvcc ||
      (input->ref && (input->ref->is_mvcc || input->ref->has_dictionary_without_values))) return HY_OK;
  hipStream_t stream = current_stream();
  const uint32_t n_chunks = input->n_chunks;
  std::vector<uint64_t> bases(size_t{n_chunks} + 1, 0);
  for (uint32_t c = 0; c < n_chunks; ++c) bases[c + 1] = (bases[c] + input->host_segments[c].size + 3) & ~uint64_t{3};   // (4 int32 = 16 bytes)
  const uint64_t padded = bases[n_chunks];
  HY_TRY(out.bases.alloc(8 * (size_t{n_chunks} + 1)));
  HY_TRY(out.values.alloc(4 * padded + 64));
  HY_TRY(out.nulls.alloc(padded + 64));
  HY_HIP(hipMemcpyAsync(out.bases.ptr, bases.data(), 8 * (size_t{n_chunks} + 1), hipMemcpyHostToDevice, stream));   // (pageable memory: copied before the call returns)
  HY_HIP(hipMemsetAsync(out.nulls.ptr, 0, padded + 64, stream));   // (the padding between chunks, and the flag behind them)
  uint32_t* found = reinterpret_cast<uint32_t*>(out.nulls.as<uint8_t>() + ((padded + 3) & ~uint64_t{3}) + 16);
  HY_TRY(export_column_at(input, out.values.ptr, out.nulls.as<uint8_t>(), out.bases.as<uint64_t>()));
  hipLaunchKernelGGL(join_any_null_byte, dim3(static_cast<uint32_t>(std::min<uint64_t>((padded + 255) / 256, 2048))), dim3(256), 0, stream, out.nulls.as<uint8_t>(), padded, found);
  uint32_t any = 0;
  HY_HIP(hipMemcpyAsync(&any, found, 4, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  if (any) return HY_OK;   // (NULL keys: the reference column itself, through the generic decoders)
  std::vector<hy_segment> segments(n_chunks ? n_chunks : 1);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hy_segment& s = segments[c];
    std::memset(&s, 0, sizeof(s));
    s.encoding = HY_ENC_UNENCODED;
    s.data_type = HY_TYPE_INT;
    s.size = input->host_segments[c].size;
    s.width = 4;
    s.data = out.values.as<int32_t>() + bases[c];
    s.ref_chunk_id = 0xFFFFFFFFu;
  }
  return hy_column_create(segments.data(), n_chunks, HY_MEM_DEVICE, &out.column);
}

static hy_status run_join(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result, bool count_only, uint64_t* count_out,
                          const hy_join_predicate* secondary = nullptr, uint32_t n_secondary = 0) {
  MaterialisedInput left_keys, right_keys;
  HY_TRY(materialise_reference_input(left, left_keys));
  HY_TRY(materialise_reference_input(right, right_keys));
  if (left_keys.column) left = left_keys.column;
  if (right_keys.column) right = right_keys.column;
  const uint32_t radix_bits = result ? result->radix_bits : 0;   // (the first attempt overwrites the caller's request with what it used)
  bool retry = false;
  t_join_returned_async = false;
  hy_status status = run_join_once(left, right, mode, result, count_only, count_out, secondary, n_secondary, &retry);
  t_last_join_hinted = t_last_join_hinted_attempt ? 1 : 0;
  if (status == HY_OK && retry) {   // the build column's key hint did not hold: it is dropped now, the two-pass build runs
    if (result) result->radix_bits = radix_bits;
    retry = false;
    status = run_join_once(left, right, mode, result, count_only, count_out, secondary, n_secondary, &retry);
    t_last_join_hinted = 2;
  }
  // HY_JOIN_ASYNC on a shape that ran synchronously: the status block is filled all the same (a following kernel may read it)
  if (result && !count_only && result->mem == HY_MEM_DEVICE && (result->flags & HY_JOIN_ASYNC) && result->status && !t_join_returned_async &&
      (status == HY_OK || status == HY_ERR_CAPACITY)) {
    hipLaunchKernelGGL(publish_join_status, dim3(1), dim3(1), 0, current_stream(), result->status, result->n_pairs, result->n_slices, status == HY_OK ? 1u : 0u);
  }
  return status;
}

}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_join_hash(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result) {
  if (!left || !right || !result) return fail(HY_ERR_INVALID, "hy_join_hash: null argument");
  HY_TRY(on_this_device(left, "hy_join_hash"));
  HY_TRY(on_this_device(right, "hy_join_hash"));
  HY_TRY(plain_column(left, &left));   // (run-length / bit-packed segments: the decoded twin, hy_device.hpp)
  HY_TRY(plain_column(right, &right));
  return run_join(left, right, mode, result, false, nullptr);
}

hy_status hy_join_hash_finish(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result) {
  if (!left || !right || !result) return fail(HY_ERR_INVALID, "hy_join_hash_finish: null argument");
  if (!(result->flags & HY_JOIN_ASYNC) || !result->status || result->mem != HY_MEM_DEVICE) return HY_OK;
  HY_TRY(on_this_device(left, "hy_join_hash_finish"));
  HY_TRY(on_this_device(right, "hy_join_hash_finish"));
  hipStream_t stream = current_stream();
  void* host = nullptr;
  void* device = nullptr;
  HY_TRY(pinned_staging(sizeof(hy_join_status), &host, &device));
  HY_HIP(hipMemcpyAsync(host, result->status, sizeof(hy_join_status), hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  const hy_join_status seen = *static_cast<const hy_join_status*>(host);
  if (!seen.build_confirmed) {   // the build column contradicted its key hint: nothing was written -- drop the hint, run the join again (two-pass build)
    HY_TRY(plain_column(left, &left));
    HY_TRY(plain_column(right, &right));
    const bool build_right = mode == HY_JOIN_LEFT || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE || mode == HY_JOIN_SEMI ||
                             (mode == HY_JOIN_INNER && left->rows > right->rows);   // (side selection of run_join_once)
    (build_right ? right : left)->join_hint.state.store(2, std::memory_order_release);
    const uint32_t flags = result->flags;
    result->flags = flags & ~HY_JOIN_ASYNC;   // (radix_bits: the first attempt left the value it used)
    const hy_status status = run_join(left, right, mode, result, false, nullptr);
    result->flags = flags;
    t_last_join_hinted = 2;
    if (status == HY_OK || status == HY_ERR_CAPACITY) {
      hipLaunchKernelGGL(publish_join_status, dim3(1), dim3(1), 0, stream, result->status, result->n_pairs, result->n_slices, status == HY_OK ? 1u : 0u);
      HY_HIP(hipStreamSynchronize(stream));
    }
    return status;
  }
  result->n_pairs = seen.n_pairs;
  result->n_slices = seen.n_slices;
  if (seen.error) return fail(HY_ERR_UNSUPPORTED, "a probe row matches more than 4 194 303 build rows");
  if (seen.n_slices > result->slice_capacity) return fail(HY_ERR_CAPACITY, "join produces %u output PosLists, slice capacity is %u", seen.n_slices, result->slice_capacity);
  if (seen.n_pairs > result->capacity) return fail(HY_ERR_CAPACITY, "join produces %llu pairs, capacity is %llu", static_cast<unsigned long long>(seen.n_pairs), static_cast<unsigned long long>(result->capacity));
  return HY_OK;
}

hy_status hy_join_hash_predicates(const hy_column* left, const hy_column* right, uint32_t mode, const hy_join_predicate* secondary, uint32_t n_secondary,
                                  hy_join_result* result) {
  if (!left || !right || !result || (n_secondary && !secondary)) return fail(HY_ERR_INVALID, "hy_join_hash_predicates: null argument");
  HY_TRY(on_this_device(left, "hy_join_hash_predicates"));
  HY_TRY(on_this_device(right, "hy_join_hash_predicates"));
  for (uint32_t i = 0; i < n_secondary; ++i) {
    HY_TRY(on_this_device(secondary[i].left_column, "hy_join_hash_predicates"));
    HY_TRY(on_this_device(secondary[i].right_column, "hy_join_hash_predicates"));
  }
  HY_TRY(plain_column(left, &left));
  HY_TRY(plain_column(right, &right));
  std::vector<hy_join_predicate> plain_secondary(secondary, secondary + n_secondary);
  for (hy_join_predicate& predicate : plain_secondary) {
    HY_TRY(plain_column(predicate.left_column, &predicate.left_column));
    HY_TRY(plain_column(predicate.right_column, &predicate.right_column));
  }
  return run_join(left, right, mode, result, false, nullptr, plain_secondary.data(), n_secondary);
}

hy_status hy_join_hash_count(const hy_column* left, const hy_column* right, uint32_t mode, uint64_t* n_pairs) {
  if (!left || !right || !n_pairs) return fail(HY_ERR_INVALID, "hy_join_hash_count: null argument");
  HY_TRY(on_this_device(left, "hy_join_hash_count"));
  HY_TRY(on_this_device(right, "hy_join_hash_count"));
  HY_TRY(plain_column(left, &left));
  HY_TRY(plain_column(right, &right));
  return run_join(left, right, mode, nullptr, true, n_pairs);
}

hy_status hy_join_hash_radix_bits(uint64_t build_rows, uint64_t probe_rows, uint32_t* radix_bits) {
  (void)probe_rows;
  if (!radix_bits) return fail(HY_ERR_INVALID, "hy_join_hash_radix_bits: null argument");
  *radix_bits = calculate_radix_bits(build_rows);
  return HY_OK;
}

// debug / tests only: 0 = the last join of this thread probed the sorted directory, 1 = a rank table, 2 = a rank table whose ranks are row numbers
// debug only: 1 if rt_probe_emit ranks with lane-ordered LDS atomics on this device, 2 if the probe said no, 0 before the first join
int hy_debug_join_lane_ordered_atomics() { return g_lds_atomic_order.load(); }

int hy_debug_join_used_rank_table(void) { return t_last_join_used_rank_table; }

// debug / tests only: see t_last_join_hinted
int hy_debug_join_build_was_hinted(void) { return t_last_join_hinted; }

// debug / tests only: 1 = the last join of this thread ran pk_count / pk_scan / pk_emit / pk_cuts (join_pkfk.hpp), 2 = pk_count_lds / pk_emit<., true>
int hy_debug_join_used_pkfk(void) { return t_last_join_used_pkfk; }

// debug only (HY_JOIN_TRACE): the per-tile phase stamps of the last join's probe_emit; not part of the public header
int hy_debug_join_trace(uint64_t* out, uint32_t capacity_tiles) {
  if (!g_join_trace) return 0;
  const uint32_t n = g_join_trace_tiles < capacity_tiles ? g_join_trace_tiles : capacity_tiles;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(out, g_join_trace, size_t{n} * 48, hipMemcpyDeviceToHost);
  return static_cast<int>(n);
}

}  // extern "C"
