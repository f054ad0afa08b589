// join_star.hpp -- a star join's probes fused into ONE pass over the fact table (included by join.hip, inside namespace hy).
//
// What it replaces (reference, CPU): the chain of JoinHash operators of a star query -- fact JOIN dim_1 ... JOIN dim_k, every
// dimension's (filtered) primary key the build side, the join result so far the probe side: materialize / partition / build / probe per
// join, operators/join_hash/join_hash_steps.hpp:274-792 -- for hy_star_join_aggregate (plan.hip), BASELINE.json configs[4].
// As an operator chain every join writes its pairs, and the carried RowIDs of every table are gathered through them for the next
// join: 39 % of an SSB query's device time was materialising intermediates that a star probe never needs (profiles/r04_ssb_kernel_stats.txt).
// Here:
//   star_column_extent / star_dim_fill   per dimension: smallest / largest key of its key column (once per column: remembered), then a direct table over that range --
//                                     one presence bit and one packed RowID (chunk << 16 | offset) per key value; a key met twice is
//                                     reported (not a primary key: the caller runs the operator chain)
//   star_probe_mask                   persistent 1024-thread workgroups, the dimensions' presence bits staged in LDS where they fit
//                                     (144 KB: SSB SF30's part + supplier + date, or customer + supplier + date); a fact row's foreign
//                                     keys arrive with 16-byte loads (eight consecutive rows per lane), a row survives if every
//                                     dimension has its key; dimensions whose bits did not fit are asked in global memory, for the rows
//                                     that survived the others only.  Leaves one byte per eight rows and a count per 8192-row tile.
//   star_scan_counts                  where every tile's survivors go
//   star_emit_rows                    the survivors' RowIDs, in FACT-TABLE ROW ORDER: the fact row's, and per dimension the RowID its
//                                     table holds for the row's key (only survivors' keys are read again)
// HBM traffic: every foreign-key column once + 1 bit per row + 8 bytes x (1 + dimensions) per surviving row.
#pragma once

constexpr uint32_t STAR_THREADS = 1024;                 // eight consecutive rows of a tile per thread
constexpr uint32_t STAR_LDS_DIMENSIONS = 4;             // dimensions whose bits a workgroup stages, at most
constexpr uint32_t STAR_LDS_WORDS = 36 * 1024;          // 144 KB of presence bits per workgroup
constexpr uint64_t STAR_MAX_RANGE = (1ull << 26) - 1;   // key values a dimension's direct table may span (8 MB of bits, 256 MB of RowIDs)
static_assert(SLICE_ROWS == STAR_THREADS * 8, "a tile is a slice: eight rows per thread");

struct StarTable {   // one dimension as the probe kernels see it
  const SliceView* views;   // the fact table's foreign-key column, slice by slice
  const uint32_t* bits;     // [words] presence bits of the key values key_min .. key_min + range
  const uint32_t* ids;      // [range + 1] packed RowID of the dimension row with that key
  uint32_t key_min, range;
  uint32_t words;
  uint32_t lds_word;        // first word of its bits in a workgroup's LDS; 0xFFFFFFFF: asked in global memory
};

struct StarArgs {
  StarTable table[HY_MAX_STAR_DIMENSIONS];   // the LDS-resident dimensions first
  uint32_t n_tables, n_lds;
  uint32_t n_tiles;
  uint8_t* masks;                            // [n_tiles][1024] bit j of byte t: row 8 t + j of the tile survived
  uint32_t* counts;                          // [n_tiles]
  const uint64_t* base;                      // [n_tiles + 1] star_emit_rows: survivors of earlier tiles
  hy_row_id* fact_rows;
  hy_row_id* table_rows[HY_MAX_STAR_DIMENSIONS];   // nullptr: nobody reads that dimension's rows
};

// Every row of a data table, chunk by chunk: the "filtered rows" of a dimension without a filter.
__global__ __launch_bounds__(256) void star_all_rows(const uint64_t* row_base, uint32_t n_chunks, hy_row_id* rows) {
  const uint32_t chunk = blockIdx.x;
  const uint64_t begin = row_base[chunk], end = row_base[chunk + 1];
  for (uint64_t i = begin + threadIdx.x; i < end; i += 256) rows[i] = hy_row_id{chunk, static_cast<uint32_t>(i - begin)};
}

// What the two kernels below need of every dimension (blockIdx.y = dimension: all dimensions in ONE launch each -- a launch per dimension was
// 30 us apiece for tables of a million rows and less, one after the other).
struct StarDimensionJobs {
  const DevSegment* segments[HY_MAX_STAR_DIMENSIONS];   // the key column
  const hy_row_id* rows[HY_MAX_STAR_DIMENSIONS];        // the dimension's rows that take part
  uint64_t n[HY_MAX_STAR_DIMENSIONS];
  const uint64_t* n_in_memory[HY_MAX_STAR_DIMENSIONS];  // ... or where their number stands (a scan's total that never went to the host)
  int64_t key_min[HY_MAX_STAR_DIMENSIONS];              // star_dim_fill
  uint32_t* bits[HY_MAX_STAR_DIMENSIONS];
  uint32_t* ids[HY_MAX_STAR_DIMENSIONS];
};

// extent[0] = smallest value ^ sign, extent[1] = largest value ^ sign of a data column (NULLs skipped); one workgroup per chunk
__global__ __launch_bounds__(256) void star_column_extent(const DevSegment* segments, unsigned long long* extent) {
  constexpr uint64_t SIGN = 1ull << 63;
  const uint32_t chunk = blockIdx.x;
  const uint32_t size = segments[chunk].size;
  uint64_t low = ~0ull, high = 0;
  for (uint32_t row = threadIdx.x; row < size; row += 256) {
    const Value v = column_value(segments, chunk, row);
    if (v.is_null) continue;
    const uint64_t biased = static_cast<uint64_t>(v.i) ^ SIGN;
    low = biased < low ? biased : low;
    high = biased > high ? biased : high;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const uint64_t other_low = __shfl_xor(low, s, 64), other_high = __shfl_xor(high, s, 64);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
  }
  __shared__ uint64_t s_low[4], s_high[4];
  if ((threadIdx.x & 63) == 0) { s_low[threadIdx.x >> 6] = low; s_high[threadIdx.x >> 6] = high; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < 4; ++w) { low = s_low[w] < low ? s_low[w] : low; high = s_high[w] > high ? s_high[w] : high; }
    if (low <= high) { atomicMin(extent, static_cast<unsigned long long>(low)); atomicMax(extent + 1, static_cast<unsigned long long>(high)); }
  }
}

// The dimensions' direct tables: bit and packed RowID of every key; *duplicate = 1 if a key comes twice.  (A dimension without a table
// -- no rows -- has n = 0.)
__global__ __launch_bounds__(256) void star_dim_fill(StarDimensionJobs jobs, uint32_t* duplicate) {
  const uint32_t d = blockIdx.y;
  const DevSegment* segments = jobs.segments[d];
  const hy_row_id* rows = jobs.rows[d];
  const uint64_t n = jobs.n_in_memory[d] ? *jobs.n_in_memory[d] : jobs.n[d];
  const int64_t key_min = jobs.key_min[d];
  uint32_t* bits = jobs.bits[d];
  uint32_t* ids = jobs.ids[d];
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) {
    const hy_row_id row = rows[i];
    const Value v = column_value(segments, row.chunk_id, row.chunk_offset);
    if (v.is_null) continue;
    const uint32_t rel = static_cast<uint32_t>(v.i - key_min);
    const uint32_t bit = 1u << (rel & 31);
    if (atomicOr(&bits[rel >> 5], bit) & bit) *duplicate = 1;
    ids[rel] = row.chunk_id << 16 | row.chunk_offset;
  }
}

// The eight consecutive stored words of a lane (rows first .. first + 7 of the slice; `first` a multiple of eight)
__device__ __forceinline__ void star_load_words(const SliceView& view, uint32_t first, u32x4_t (&words)[2], uint32_t* bias) {
  const char* base = static_cast<const char*>(view.data);
  const uint32_t row = view.row_begin + (first < view.row_count ? first : 0);   // (a lane without rows reads the slice's first words)
  if (view.kind == VIEW_FOR8) load_batch_words<1>(base, row, words);
  else if (view.kind == VIEW_FOR16) load_batch_words<2>(base, row, words);
  else load_batch_words<4>(base, row, words);
  *bias = view.kind == VIEW_INT32 ? 0u : static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[row / HY_FOR_BLOCK_SIZE]);
}

// Which of the lane's eight rows have their key in the table (bits: the table's presence words, LDS)?
template <uint32_t WIDTH>
__device__ __forceinline__ uint32_t star_test_words(const u32x4_t (&words)[2], uint32_t bias, const StarTable& table, const uint32_t* bits) {
  uint32_t found = 0;
  const uint32_t delta = bias - table.key_min;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t rel = batch_word<WIDTH>(words, j) + delta;   // (32-bit: the wrapped difference of two int32 values is their distance, or larger than any range)
    const uint32_t word = bits[rel <= table.range ? rel >> 5 : 0u];
    found |= (rel <= table.range ? (word >> (rel & 31)) & 1u : 0u) << j;
  }
  return found;
}

__global__ __launch_bounds__(STAR_THREADS) void star_probe_mask(StarArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_star_bits[];
  __shared__ uint32_t s_count;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  for (uint32_t d = 0; d < a.n_lds; ++d) {
    const StarTable& table = a.table[d];
    for (uint32_t i = tid; i < table.words; i += STAR_THREADS) s_star_bits[table.lds_word + i] = table.bits[i];
  }
  const uint32_t first = tid * 8;
  for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    if (tid == 0) s_count = 0;
    __syncthreads();   // (the bits are staged; the count of the tile before has been written)
    // the tile's stored words of every LDS-resident dimension's foreign key, all requested before the first is looked at
    u32x4_t words[STAR_LDS_DIMENSIONS][2];
    uint32_t bias[STAR_LDS_DIMENSIONS], kind[STAR_LDS_DIMENSIONS];
    uint32_t rows = 0;
#pragma unroll
    for (uint32_t d = 0; d < STAR_LDS_DIMENSIONS; ++d) {
      words[d][0] = words[d][1] = u32x4_t{0, 0, 0, 0};
      bias[d] = kind[d] = 0;
      if (d < a.n_lds) {
        const SliceView view = a.table[d].views[tile];
        star_load_words(view, first, words[d], &bias[d]);
        kind[d] = view.kind;
        rows = view.row_count;
      }
    }
    if (a.n_lds == 0) rows = a.table[0].views[tile].row_count;
    uint32_t alive = first >= rows ? 0u : (rows - first < 8 ? (1u << (rows - first)) - 1u : 0xFFu);
#pragma unroll
    for (uint32_t d = 0; d < STAR_LDS_DIMENSIONS; ++d) {
      if (d >= a.n_lds) continue;
      const uint32_t* bits = s_star_bits + a.table[d].lds_word;
      alive &= kind[d] == VIEW_FOR8 ? star_test_words<1>(words[d], bias[d], a.table[d], bits) : kind[d] == VIEW_FOR16 ? star_test_words<2>(words[d], bias[d], a.table[d], bits)
                                                                                                                     : star_test_words<4>(words[d], bias[d], a.table[d], bits);
    }
    // the dimensions whose bits did not fit: asked in global memory, the rows that are still alive only
    for (uint32_t d = a.n_lds; d < a.n_tables; ++d) {
      if (!__any(alive != 0)) break;
      const StarTable& table = a.table[d];
      const SliceView view = table.views[tile];
      uint32_t pending = alive;
      while (pending) {
        const uint32_t j = __ffs(pending) - 1;
        pending &= pending - 1;
        const uint32_t rel = static_cast<uint32_t>(view_key(view, view.row_begin + first + j)) - table.key_min;
        if (rel > table.range || !((table.bits[rel >> 5] >> (rel & 31)) & 1u)) alive &= ~(1u << j);
      }
    }
    a.masks[static_cast<size_t>(tile) * STAR_THREADS + tid] = static_cast<uint8_t>(alive);
    uint32_t survivors = __popc(alive);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) survivors += __shfl_xor(survivors, d, 64);
    if (lane == 0 && survivors) atomicAdd(&s_count, survivors);
    __syncthreads();
    if (tid == 0) a.counts[tile] = s_count;
  }
}

// Exclusive prefix of the tiles' survivor counts (n is tens of thousands): base[i] = survivors of the tiles before i, base[n] = all of them.
// Blocks of 16 384 counts go through LDS: coalesced loads and stores, sixteen consecutive counts per thread in between.
__global__ __launch_bounds__(1024) void star_scan_counts(const uint32_t* counts, uint64_t* base, uint32_t n) {
  constexpr uint32_t PER = 16, BLOCK = 1024 * PER;
  __shared__ uint32_t s_counts[BLOCK + BLOCK / PER];   // (element i at i + i / 16: a thread's sixteen start seventeen words apart)
  __shared__ uint64_t s_wave[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t carry = 0;
  for (uint32_t begin = 0; begin < n; begin += BLOCK) {
    const uint32_t m = n - begin < BLOCK ? n - begin : BLOCK;
    for (uint32_t i = tid; i < BLOCK; i += 1024) s_counts[i + i / PER] = i < m ? counts[begin + i] : 0u;
    __syncthreads();
    uint32_t* mine = s_counts + tid * (PER + 1);
    uint64_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) sum += mine[k];
    uint64_t inclusive = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t other = __shfl_up(inclusive, d, 64);
      if (lane >= static_cast<uint32_t>(d)) inclusive += other;
    }
    if (lane == 63) s_wave[wave] = inclusive;
    __syncthreads();
    uint64_t run = inclusive - sum, total = 0;   // (relative to the block: a block holds 16 384 tiles of at most 8192 survivors, 32 bits)
    for (uint32_t w = 0; w < 16; ++w) { if (w < wave) run += s_wave[w]; total += s_wave[w]; }
    // the prefixes go back into the thread's sixteen LDS words and leave coalesced (a thread that stores its own sixteen 64-bit words writes a
    // line of its own per instruction: 16 000 transactions from one CU were most of this kernel's 25 us)
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { const uint32_t count = mine[k]; mine[k] = static_cast<uint32_t>(run); run += count; }
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 1024) base[begin + i] = carry + s_counts[i + i / PER];
    carry += total;
    __syncthreads();
  }
  if (tid == 0) base[n] = carry;
}

// The survivors' RowIDs.  One workgroup per STAR_EMIT_TILES consecutive tiles: the set bits of their masks become a list of (tile, row)
// in LDS -- in row order, a thread owns 256 consecutive rows -- and the list is worked off one survivor per thread: the output stores of a
// wave are consecutive, and the dependent loads of a survivor (its stored word and block minimum per dimension, the dimension's RowID)
// are in flight for a whole list at once instead of one tile's handful per workgroup (22 000 workgroups of a few dozen survivors each
// took 145 us at SSB SF30).
constexpr uint32_t STAR_EMIT_TILES = 8;
constexpr uint32_t STAR_EMIT_LIST = 8192;   // survivors per pass over a workgroup's tiles
__global__ __launch_bounds__(256) void star_emit_rows(StarArgs a) {
  __shared__ uint32_t s_list[STAR_EMIT_LIST];   // tile in group << 13 | row in tile
  __shared__ uint32_t s_wave[4];
  __shared__ SliceView s_views[STAR_EMIT_TILES][HY_MAX_STAR_DIMENSIONS];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t first_tile = blockIdx.x * STAR_EMIT_TILES;
  const uint32_t n_tiles = a.n_tiles - first_tile < STAR_EMIT_TILES ? a.n_tiles - first_tile : STAR_EMIT_TILES;
  const uint64_t group_base = a.base[first_tile];
  const uint32_t total = static_cast<uint32_t>(a.base[first_tile + n_tiles] - group_base);
  if (total == 0) return;   // (uniform)
  for (uint32_t i = tid; i < n_tiles * a.n_tables; i += 256) s_views[i / a.n_tables][i % a.n_tables] = a.table[i % a.n_tables].views[first_tile + i / a.n_tables];
  // the thread's 256 consecutive rows: eight mask words
  const uint32_t* words = reinterpret_cast<const uint32_t*>(a.masks + static_cast<size_t>(first_tile) * STAR_THREADS);
  uint32_t mask[8], mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    const uint32_t w = tid * 8 + k;
    mask[k] = w < n_tiles * (STAR_THREADS / 4) ? words[w] : 0u;
    mine += __popc(mask[k]);
  }
  uint32_t inclusive = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t other = __shfl_up(inclusive, d, 64);
    if (lane >= static_cast<uint32_t>(d)) inclusive += other;
  }
  if (lane == 63) s_wave[wave] = inclusive;
  __syncthreads();
  uint32_t before = inclusive - mine;
  for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
  for (uint32_t pass_begin = 0; pass_begin < total; pass_begin += STAR_EMIT_LIST) {
    if (pass_begin) __syncthreads();   // (the list of the pass before has been worked off)
    uint32_t rank = before;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
      uint32_t bits = mask[k];
      const uint32_t row0 = (tid * 8 + k) * 32;   // row of bit 0, counted from the group's first row: tile = row0 / 8192
      while (bits) {
        const uint32_t j = __ffs(bits) - 1;
        bits &= bits - 1;
        if (rank - pass_begin < STAR_EMIT_LIST) s_list[rank - pass_begin] = row0 + j;   // (unsigned: ranks of earlier passes wrap to huge values)
        ++rank;
      }
    }
    __syncthreads();
    const uint32_t in_pass = total - pass_begin < STAR_EMIT_LIST ? total - pass_begin : STAR_EMIT_LIST;
    // eight survivors per thread at a time, table by table: their eight stored words are requested together, then their eight RowIDs, then
    // the eight stores leave -- loads and stores share one counter on this hardware, a loop that stores after every load waits for each store
    constexpr uint32_t AT_ONCE = 8;
    for (uint32_t first = 0; first < in_pass; first += 256 * AT_ONCE) {
      uint32_t tile[AT_ONCE], row[AT_ONCE];
      bool there[AT_ONCE];
#pragma unroll
      for (uint32_t k = 0; k < AT_ONCE; ++k) {
        const uint32_t i = first + k * 256 + tid;
        there[k] = i < in_pass;
        const uint32_t entry = s_list[there[k] ? i : 0u];
        tile[k] = entry >> 13;
        row[k] = entry & 8191u;
      }
      const uint64_t out = group_base + pass_begin + first + tid;
#pragma unroll
      for (uint32_t k = 0; k < AT_ONCE; ++k) {
        const SliceView& fact = s_views[tile[k]][0];
        if (there[k]) a.fact_rows[out + k * 256] = hy_row_id{fact.chunk, fact.row_begin + row[k]};
      }
      for (uint32_t d = 0; d < a.n_tables; ++d) {
        if (!a.table_rows[d]) continue;
        uint32_t key[AT_ONCE], id[AT_ONCE];
#pragma unroll
        for (uint32_t k = 0; k < AT_ONCE; ++k) {
          const SliceView& view = s_views[tile[k]][d];
          key[k] = static_cast<uint32_t>(view_key(view, view.row_begin + row[k]));   // (a thread without a survivor reads the list's first once more)
        }
#pragma unroll
        for (uint32_t k = 0; k < AT_ONCE; ++k) id[k] = a.table[d].ids[key[k] - a.table[d].key_min];
#pragma unroll
        for (uint32_t k = 0; k < AT_ONCE; ++k) {
          if (there[k]) a.table_rows[d][out + k * 256] = hy_row_id{id[k] >> 16, id[k] & 0xFFFFu};
        }
      }
    }
  }
}

// Can the probe kernels read this column as a fact table's foreign key (int32 keys, every segment one a SliceView describes)?
static bool star_reads_fact_key(const hy_column* column) {
  if (!column || column->is_reference || column->is_mvcc || column->has_compressed || column->data_type != HY_TYPE_INT) return false;
  for (const hy_segment& s : column->host_segments) {
    if (reinterpret_cast<uintptr_t>(s.data) % 16 != 0 || s.nulls) return false;
    if (s.size == 0 || !s.data) return false;   // (a tile of an empty chunk would load from its null buffers)
    if (!((s.encoding == HY_ENC_UNENCODED && s.data_type == HY_TYPE_INT) || (s.encoding == HY_ENC_FRAME_OF_REFERENCE && (s.width == 1 || s.width == 2 || s.width == 4)))) return false;
  }
  return true;
}

// star_probe_rows (hy_device.hpp): see the top of this file.  *applicable = false: nothing was produced, the caller joins dimension by dimension.
hy_status star_probe_rows(const StarProbeDimension* dimensions, uint32_t n_dimensions, DeviceBuffer& fact_rows, std::vector<std::unique_ptr<DeviceBuffer>>& dimension_rows, uint64_t* n_rows,
                          bool* applicable) {
  *applicable = false;
  *n_rows = 0;
  if (!option(HY_OPT_STAR_FUSED_PROBE) || n_dimensions == 0 || n_dimensions > HY_MAX_STAR_DIMENSIONS) return HY_OK;
  const hy_column* shape = dimensions[0].fact_key;
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    const StarProbeDimension& dim = dimensions[d];
    if (!star_reads_fact_key(dim.fact_key) || !dim.key || dim.key->data_type != HY_TYPE_INT || dim.key->is_mvcc) return HY_OK;
    // the dimension's keys are read cell by cell through column_value: RunLength segments, bit-packed vectors and dictionaries whose values
    // are not on the device are not decoded there -- such a key column keeps the join-by-join chain (hy_join_hash reads its decoded twin)
    const hy_column* key_data = dim.key->is_reference ? dim.key->ref : dim.key;
    if (dim.key->has_compressed || dim.key->has_dictionary_without_values || (key_data && (key_data->has_compressed || key_data->has_dictionary_without_values))) return HY_OK;
    if (dim.fact_key->n_chunks != shape->n_chunks || dim.fact_key->n_slices != shape->n_slices || dim.fact_key->rows != shape->rows) return HY_OK;
    for (uint32_t c = 0; c < shape->n_chunks; ++c) if (dim.fact_key->host_segments[c].size != shape->host_segments[c].size) return HY_OK;
    // a dimension row's RowID is packed into 32 bits
    const hy_column* rows_of = dim.key->is_reference ? dim.key->ref : dim.key;
    if (!rows_of || rows_of->n_chunks > 65536) return HY_OK;
    for (const hy_segment& s : rows_of->host_segments) if (s.size > 65536) return HY_OK;
  }
  hipStream_t stream = current_stream();
  // ---- the dimensions' key ranges: the extent of the whole key column (a superset of its filtered rows' keys), remembered by the column --
  // one look at the keys and one host read the first time a column serves as a dimension key, none afterwards ------------------------------
  std::vector<uint64_t> extent(2 * size_t{n_dimensions});
  constexpr uint64_t SIGN = 1ull << 63;
  {
    DeviceBuffer extents;
    std::vector<uint32_t> asked;
    for (uint32_t d = 0; d < n_dimensions; ++d) if (dimensions[d].key->extent_state.load(std::memory_order_acquire) == 0) asked.push_back(d);
    if (!asked.empty()) {
      HY_TRY(extents.alloc(16 * asked.size() + 64));
      std::vector<uint64_t> initial(2 * asked.size());
      for (size_t i = 0; i < asked.size(); ++i) { initial[2 * i] = ~0ull; initial[2 * i + 1] = 0; }
      HY_HIP(hipMemcpyAsync(extents.ptr, initial.data(), 16 * asked.size(), hipMemcpyHostToDevice, stream));
      for (size_t i = 0; i < asked.size(); ++i) {
        const hy_column* key = dimensions[asked[i]].key;
        if (key->n_chunks) hipLaunchKernelGGL(star_column_extent, dim3(key->n_chunks), dim3(256), 0, stream, key->d_segments, extents.as<unsigned long long>() + 2 * i);
      }
      std::vector<uint64_t> found(2 * asked.size());
      HY_HIP(hipMemcpyAsync(found.data(), extents.ptr, 16 * asked.size(), hipMemcpyDeviceToHost, stream));
      HY_HIP(hipStreamSynchronize(stream));
      for (size_t i = 0; i < asked.size(); ++i) {
        const hy_column* key = dimensions[asked[i]].key;
        const bool any = found[2 * i] <= found[2 * i + 1];
        key->extent_min.store(any ? static_cast<int64_t>(found[2 * i] ^ SIGN) : 0, std::memory_order_relaxed);
        key->extent_max.store(any ? static_cast<int64_t>(found[2 * i + 1] ^ SIGN) : 0, std::memory_order_relaxed);
        key->extent_state.store(any ? 1u : 2u, std::memory_order_release);
      }
    }
    for (uint32_t d = 0; d < n_dimensions; ++d) {
      const hy_column* key = dimensions[d].key;
      const bool any = key->extent_state.load(std::memory_order_acquire) == 1 && dimensions[d].n_rows != 0;
      extent[2 * d] = any ? static_cast<uint64_t>(key->extent_min.load(std::memory_order_relaxed)) ^ SIGN : ~0ull;
      extent[2 * d + 1] = any ? static_cast<uint64_t>(key->extent_max.load(std::memory_order_relaxed)) ^ SIGN : 0;
    }
  }
  StarDimensionJobs jobs;
  std::memset(&jobs, 0, sizeof(jobs));
  uint64_t most_rows = 0;
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    jobs.segments[d] = dimensions[d].key->d_segments;
    jobs.rows[d] = dimensions[d].rows;
    jobs.n[d] = dimensions[d].n_rows;
    jobs.n_in_memory[d] = dimensions[d].d_n_rows;
    most_rows = std::max(most_rows, dimensions[d].n_rows);
  }
  const uint32_t job_grid = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((most_rows + 1023) / 1024, 256)));
  // ---- the direct tables -------------------------------------------------------------------------------------------------------
  struct Built { DeviceBuffer bits, ids; int64_t key_min = 0; uint64_t range = 0; uint32_t words = 0; bool empty = false; };
  std::vector<std::unique_ptr<Built>> built(n_dimensions);
  DeviceBuffer duplicate;
  HY_TRY(duplicate.alloc(64));
  HY_HIP(hipMemsetAsync(duplicate.ptr, 0, 4, stream));
  bool nothing_joins = false;
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    built[d] = std::make_unique<Built>();
    Built& b = *built[d];
    if (extent[2 * d] > extent[2 * d + 1]) { b.empty = true; nothing_joins = true; continue; }   // no row (or only NULL keys): an Inner join with it is empty
    const int64_t low = static_cast<int64_t>(extent[2 * d] ^ SIGN), high = static_cast<int64_t>(extent[2 * d + 1] ^ SIGN);
    if (low < INT32_MIN || high > INT32_MAX || static_cast<uint64_t>(high - low) > STAR_MAX_RANGE) return HY_OK;   // (a sparse key: the rank table / directory of hy_join_hash)
    b.key_min = low;
    b.range = static_cast<uint64_t>(high - low);
    b.words = static_cast<uint32_t>(b.range >> 5) + 1;
    HY_TRY(b.bits.alloc(4 * size_t{b.words} + 16));
    HY_TRY(b.ids.alloc(4 * (b.range + 1) + 16));
    HY_HIP(hipMemsetAsync(b.bits.ptr, 0, 4 * size_t{b.words}, stream));
    jobs.key_min[d] = b.key_min;
    jobs.bits[d] = b.bits.as<uint32_t>();
    jobs.ids[d] = b.ids.as<uint32_t>();
  }
  for (uint32_t d = 0; d < n_dimensions; ++d) if (built[d]->empty) { jobs.n[d] = 0; jobs.n_in_memory[d] = nullptr; }
  if (!nothing_joins) hipLaunchKernelGGL(star_dim_fill, dim3(job_grid, n_dimensions), dim3(256), 0, stream, jobs, duplicate.as<uint32_t>());
  dimension_rows.clear();
  dimension_rows.resize(n_dimensions);
  if (nothing_joins) {   // (still "applicable": the join result is empty)
    HY_TRY(fact_rows.alloc(sizeof(hy_row_id)));
    for (uint32_t d = 0; d < n_dimensions; ++d) { dimension_rows[d] = std::make_unique<DeviceBuffer>(); HY_TRY(dimension_rows[d]->alloc(sizeof(hy_row_id))); }
    *applicable = true;
    return HY_OK;
  }
  // ---- which dimensions' bits go to LDS: the smallest tables first, while they fit ----------------------------------------------
  std::vector<uint32_t> order(n_dimensions);
  for (uint32_t d = 0; d < n_dimensions; ++d) order[d] = d;
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return built[x]->words != built[y]->words ? built[x]->words < built[y]->words : x < y; });
  StarArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_tables = n_dimensions;
  uint32_t lds_words = 0;
  std::vector<uint32_t> slot_of(n_dimensions, 0);
  {
    std::vector<uint32_t> in_lds, in_memory;
    for (uint32_t d : order) {
      const uint32_t words = (built[d]->words + 3) & ~3u;
      if (in_lds.size() < STAR_LDS_DIMENSIONS && lds_words + words <= STAR_LDS_WORDS) { in_lds.push_back(d); lds_words += words; }
      else in_memory.push_back(d);
    }
    // (asked in memory: the most selective first -- fewest keys per key value -- so that the others see fewer rows)
    std::sort(in_memory.begin(), in_memory.end(), [&](uint32_t x, uint32_t y) {
      return static_cast<double>(dimensions[x].n_rows) / static_cast<double>(built[x]->range + 1) < static_cast<double>(dimensions[y].n_rows) / static_cast<double>(built[y]->range + 1);
    });
    uint32_t slot = 0, at = 0;
    for (uint32_t d : in_lds) { slot_of[d] = slot; a.table[slot].lds_word = at; at += (built[d]->words + 3) & ~3u; ++slot; }
    a.n_lds = slot;
    for (uint32_t d : in_memory) { slot_of[d] = slot; a.table[slot].lds_word = 0xFFFFFFFFu; ++slot; }
  }
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    StarTable& t = a.table[slot_of[d]];
    t.views = dimensions[d].fact_key->d_slice_views;
    t.bits = built[d]->bits.as<uint32_t>();
    t.ids = built[d]->ids.as<uint32_t>();
    t.key_min = static_cast<uint32_t>(static_cast<int32_t>(built[d]->key_min));
    t.range = static_cast<uint32_t>(built[d]->range);
    t.words = built[d]->words;
  }
  a.n_tiles = shape->n_slices;
  DeviceBuffer masks, counts, base;
  HY_TRY(masks.alloc(size_t{a.n_tiles} * STAR_THREADS + 16));
  HY_TRY(counts.alloc(4 * (size_t{a.n_tiles} + 1)));
  HY_TRY(base.alloc(8 * (size_t{a.n_tiles} + 2)));
  a.masks = masks.as<uint8_t>();
  a.counts = counts.as<uint32_t>();
  a.base = base.as<uint64_t>();
  static OncePerDevice lds_raised;
  uint64_t device_bit = 0;
  if (lds_raised.pending(&device_bit)) {
    HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(star_probe_mask), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAR_LDS_WORDS));
    lds_raised.done(device_bit);
  }
  const uint32_t probe_grid = std::max(1u, std::min(a.n_tiles, device_cu_count()));
  profile_begin(stream, HY_KERNEL_JOIN_PROBE);
  hipLaunchKernelGGL(star_probe_mask, dim3(probe_grid), dim3(STAR_THREADS), 4 * size_t{lds_words}, stream, a);
  profile_end(stream);
  hipLaunchKernelGGL(star_scan_counts, dim3(1), dim3(1024), 0, stream, counts.as<uint32_t>(), base.as<uint64_t>(), a.n_tiles);
  uint64_t total = 0;
  uint32_t key_twice = 0;
  HY_HIP(hipMemcpyAsync(&total, base.as<uint64_t>() + a.n_tiles, 8, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipMemcpyAsync(&key_twice, duplicate.ptr, 4, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  if (key_twice) return HY_OK;   // a dimension key that is not unique: JoinHash's business
  // ---- the survivors' RowIDs -------------------------------------------------------------------------------------------------------
  HY_TRY(fact_rows.alloc(sizeof(hy_row_id) * std::max<uint64_t>(1, total)));
  a.fact_rows = fact_rows.as<hy_row_id>();
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    if (!dimensions[d].want_rows) continue;
    dimension_rows[d] = std::make_unique<DeviceBuffer>();
    HY_TRY(dimension_rows[d]->alloc(sizeof(hy_row_id) * std::max<uint64_t>(1, total)));
    a.table_rows[slot_of[d]] = dimension_rows[d]->as<hy_row_id>();
  }
  if (total) hipLaunchKernelGGL(star_emit_rows, dim3((a.n_tiles + STAR_EMIT_TILES - 1) / STAR_EMIT_TILES), dim3(256), 0, stream, a);
  HY_HIP(hipGetLastError());
  *n_rows = total;
  *applicable = true;
  return HY_OK;
}

// Every row of a data column's table as a PosList (a dimension without a filter)
hy_status star_all_rows_of(const hy_column* column, DeviceBuffer& rows) {
  HY_TRY(rows.alloc(sizeof(hy_row_id) * std::max<uint64_t>(1, column->rows)));
  if (column->n_chunks) hipLaunchKernelGGL(star_all_rows, dim3(column->n_chunks), dim3(256), 0, current_stream(), column->d_row_base, column->n_chunks, rows.as<hy_row_id>());
  return HY_OK;
}
