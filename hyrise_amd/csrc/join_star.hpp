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
//   star_finish / star_finish_compact the plan's Projection + AggregateHash inside the join (round 6): where every GROUP BY column and
//                                     every aggregate input is an int / long column of a joined table, the survivors are grouped where
//                                     they are found -- no RowIDs written, no columns exported, no second pass by hy_aggregate_hash
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
  // rows == nullptr: EVERY row of the table, tested here against the dimension's filter (round 6: a TableScan, a prefix and a translation per
  // dimension -- three launches over a table of a few thousand to a million rows -- were a tenth of an SSB query) -- or against none
  const uint64_t* row_base[HY_MAX_STAR_DIMENSIONS];     // [n_chunks + 1] first row of every chunk of the table
  uint32_t n_chunks[HY_MAX_STAR_DIMENSIONS];
  uint32_t chunk_rows[HY_MAX_STAR_DIMENSIONS];          // rows of every chunk but the last where they agree, else 0
  const DevSegment* filter_segments[HY_MAX_STAR_DIMENSIONS];   // nullptr: no filter
  const ScanJob* filter_jobs[HY_MAX_STAR_DIMENSIONS];          // [n_chunks] the filter's normalised test per chunk (prepare_scan_jobs)
};

// One row of a data segment against its chunk's job -- what scan_slices decides for the row (hy_scan_job.hpp; JOB_RANGE does not occur:
// prepare_scan_jobs asks for none).  NULL never matches a comparison.
__device__ __forceinline__ bool star_filter_row(const DevSegment& s, const ScanJob& job, uint32_t row) {
  if (job.mode == JOB_ALL) return true;
  if (job.mode == JOB_NONE) return false;
  const bool invert = job.flags & JF_INVERT;
  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = aload_compressed(s.data, s.width, row);
    if (job.kind == KIND_VALUE_ID_SET) return vid < job.null_vid && ((reinterpret_cast<const uint64_t*>(job.lo)[vid >> 6] >> (vid & 63)) & 1) != 0;
    if (job.kind == KIND_NULLTEST) return (vid == s.aux_size) != invert;
    const bool in = (vid - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
    return (in != invert) && vid != job.null_vid;
  }
  const bool is_null = s.nulls ? ((s.nulls[row >> 6] >> (row & 63)) & 1) != 0 : false;
  if (job.kind == KIND_NULLTEST) return is_null != invert;
  if (is_null) return false;
  bool in;
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    const uint32_t x = aload_compressed(s.data, s.width, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]);
    in = (x - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
  } else if (job.kind == KIND_U32) {
    in = (static_cast<const uint32_t*>(s.data)[row] - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
  } else if (job.kind == KIND_I64) {
    in = (static_cast<const uint64_t*>(s.data)[row] - job.lo) <= job.span;
  } else if (job.kind == KIND_F32) {
    const float x = static_cast<const float*>(s.data)[row];
    const float lower = __uint_as_float(static_cast<uint32_t>(job.lo)), upper = __uint_as_float(static_cast<uint32_t>(job.span));
    in = ((job.flags & JF_LOWER_INCL) ? x >= lower : x > lower) && ((job.flags & JF_UPPER_INCL) ? x <= upper : x < upper);
  } else {
    const double x = static_cast<const double*>(s.data)[row];
    const double lower = __longlong_as_double(static_cast<long long>(job.lo)), upper = __longlong_as_double(static_cast<long long>(job.span));
    in = ((job.flags & JF_LOWER_INCL) ? x >= lower : x > lower) && ((job.flags & JF_UPPER_INCL) ? x <= upper : x < upper);
  }
  return in != invert;
}

// Row i of a table (rows counted through its chunks) -> RowID: the chunk whose first row is the last at or below i
__device__ __forceinline__ hy_row_id star_row_of_table(const uint64_t* row_base, uint32_t n_chunks, uint32_t chunk_rows, uint64_t i) {
  if (chunk_rows) { const uint32_t chunk = static_cast<uint32_t>(i / chunk_rows); return hy_row_id{chunk, static_cast<uint32_t>(i - uint64_t{chunk} * chunk_rows)}; }   // (chunks of one size, the last may be shorter)
  uint32_t low = 0, high = n_chunks;   // row_base[low] <= i < row_base[high]
  while (high - low > 1) {
    const uint32_t middle = (low + high) / 2;
    if (row_base[middle] <= i) low = middle; else high = middle;
  }
  return hy_row_id{low, static_cast<uint32_t>(i - row_base[low])};
}

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
  const uint32_t d = blockIdx.y, lane = threadIdx.x & 63;
  const DevSegment* segments = jobs.segments[d];
  const hy_row_id* rows = jobs.rows[d];
  const uint64_t n = jobs.n_in_memory[d] ? *jobs.n_in_memory[d] : jobs.n[d];
  const int64_t key_min = jobs.key_min[d];
  uint32_t* bits = jobs.bits[d];
  uint32_t* ids = jobs.ids[d];
  // (the loop's trip count is the wave's: the lanes of a wave merge their bits before they go to memory -- a dimension's rows come in key order
  //  more often than not, 64 of them meet in two or three words, and an atomic per row on those words was most of this kernel)
  for (uint64_t first = (static_cast<uint64_t>(blockIdx.x) * 256 + (threadIdx.x & ~63u)); first < n; first += static_cast<uint64_t>(gridDim.x) * 256) {
    const uint64_t i = first + lane;
    bool valid = i < n;
    uint32_t rel = 0;
    if (valid) {
      const hy_row_id row = rows ? rows[i] : star_row_of_table(jobs.row_base[d], jobs.n_chunks[d], jobs.chunk_rows[d], i);
      if (!rows && jobs.filter_segments[d]) valid = star_filter_row(jobs.filter_segments[d][row.chunk_id], jobs.filter_jobs[d][row.chunk_id], row.chunk_offset);
    if (valid) {
      const Value v = column_value(segments, row.chunk_id, row.chunk_offset);
      valid = !v.is_null;
      rel = static_cast<uint32_t>(v.i - key_min);
      if (valid) ids[rel] = row.chunk_id << 16 | row.chunk_offset;
    }
    }
    const uint32_t word = rel >> 5, bit = 1u << (rel & 31);
    uint64_t todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll(static_cast<long long>(todo)) - 1;
      const uint32_t w = static_cast<uint32_t>(__shfl(static_cast<int>(word), leader, 64));
      const bool mine = valid && word == w;
      const uint64_t same = __ballot(mine);
      uint32_t merged = mine ? bit : 0u;
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) merged |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(merged), s, 64));
      if (static_cast<int>(lane) == leader) {
        const uint32_t before = atomicOr(&bits[w], merged);
        if ((before & merged) || __popc(merged) != __popcll(same)) *duplicate = 1;   // a key another wave has set, or two lanes of this one
      }
      todo &= ~same;
      if (__popcll(same) <= 2 && todo) {   // (rows that passed a selective filter: their keys are far apart -- an atomic each, no more merging)
        if ((todo >> lane) & 1) { if (atomicOr(&bits[word], bit) & bit) *duplicate = 1; }
        break;
      }
    }
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
  // (32-bit: the wrapped difference of two int32 values is their distance, or larger than any range.  A key outside the table asks bit
  //  range + 1, which the host keeps zero -- one v_min instead of two compares and two selects per row: the kernel is bound by its
  //  instructions, 117 M vector ones per SF30 pass, profiles/r06_star_traffic.txt)
  //  Six vector instructions a lookup: unpack + add, clamp, word index (a bit-field extract, so that the compiler keeps index x 4 + base as
  //  one shift-add), the LDS read, the bit (a bit-field extract at a variable offset), shift-or into the result.
  typedef __attribute__((address_space(3))) const uint32_t lds_word;
  const uint32_t delta = bias - table.key_min, beyond = table.range + 1;
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_word*)bits));
  uint32_t rel[8], word[8];
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t distance = batch_word<WIDTH>(words, j) + delta;
    rel[j] = distance < beyond ? distance : beyond;
    uint32_t index;   // (opaque to the optimiser, which otherwise rewrites (rel >> 5) * 4 + base as shift, mask, add)
    asm("v_lshrrev_b32 %0, 5, %1" : "=v"(index) : "v"(rel[j]));
    word[j] = *(lds_word*)static_cast<uintptr_t>(index * 4u + base);
  }
  uint32_t found = 0;
#pragma unroll
  for (int j = 7; j >= 0; --j) found = (found << 1) | __builtin_amdgcn_ubfe(word[j], rel[j], 1);
  return found;
}

// A tile's stored words of every LDS-resident dimension's foreign key, requested without a branch and without a load under a condition (what
// the kernel below asks for one tile AHEAD: a load on some control-flow paths only makes the compiler wait for everything in flight where the
// paths meet -- the prefetch would be waited for where it is issued).  Words as load_batch_words leaves them.  1-byte offsets: sixteen bytes
// from eight rows back (the lane's are the upper half) or, for the slice's first rows, from the front; 2-byte offsets: the sixteen bytes; 4-byte
// words: two loads -- the narrower kinds repeat their first.  The block minimum: of the lane's first row; plain int32 values read the word at
// the front of the data instead, not used.
struct StarTileWords {
  u32x4_t words[STAR_LDS_DIMENSIONS][2];
  uint32_t bias[STAR_LDS_DIMENSIONS];
};
template <uint32_t N_WORDS>
__device__ __forceinline__ void star_request_tile(const StarArgs& a, uint32_t tile, uint32_t first, StarTileWords& t, uint32_t (&kind)[STAR_LDS_DIMENSIONS], uint32_t* rows) {
  typedef __attribute__((address_space(1))) const u32x4_t global_quad;
  typedef __attribute__((address_space(1))) const uint32_t global_word;
  *rows = uniform_view(a.table[0].views + tile).row_count;
#pragma unroll
  for (uint32_t d = 0; d < STAR_LDS_DIMENSIONS; ++d) {
    kind[d] = 0;
    if (d >= N_WORDS) continue;   // (compile time: the kernel is instantiated per number of dimensions whose words it streams -- no load is under a run-time condition)
    const SliceView view = uniform_view(a.table[d].views + tile);
    kind[d] = view.kind;
    const uint32_t width = view.kind == VIEW_FOR8 ? 1u : view.kind == VIEW_FOR16 ? 2u : 4u;
    const uint32_t row = view.row_begin + (first < view.row_count ? first : 0u);
    const uint32_t back = width == 1 && row >= 8 ? 8u : 0u;
    const char* at = static_cast<const char*>(view.data) + row * width - back;
    const u32x4_t v0 = *(global_quad*)at;
    const u32x4_t v1 = *(global_quad*)(at + (width == 4 ? 16 : 0));
    const bool biased = view.kind != VIEW_INT32;
    global_word* minima = (global_word*)(biased ? view.aux : view.data);
    t.bias[d] = minima[biased ? row / HY_FOR_BLOCK_SIZE : 0u];
    t.words[d][0] = back ? u32x4_t{v0.z, v0.w, 0, 0} : v0;
    t.words[d][1] = v1;
  }
}

// Which of the tile's rows survive every dimension -> masks[tile], counts[tile] (zeroed by the host; a wave adds its survivors: no barrier --
// the sixteen waves of the workgroup run their tiles' loads and lookups independently of each other)
template <uint32_t N_LDS, uint32_t N_STREAM>
__device__ __forceinline__ void star_probe_tile(const StarArgs& a, uint32_t tile, uint32_t tid, uint32_t first, const StarTileWords& t, const uint32_t (&kind)[STAR_LDS_DIMENSIONS], uint32_t rows,
                                                const uint32_t* s_star_bits) {
  const uint32_t lane = tid & 63;
  uint32_t alive = first >= rows ? 0u : (rows - first < 8 ? (1u << (rows - first)) - 1u : 0xFFu);
#pragma unroll
  for (uint32_t d = 0; d < STAR_LDS_DIMENSIONS; ++d) {
    if (d >= N_LDS) continue;
    const uint32_t* bits = s_star_bits + a.table[d].lds_word;
    const uint32_t bias = kind[d] == VIEW_INT32 ? 0u : t.bias[d];
    alive &= kind[d] == VIEW_FOR8 ? star_test_words<1>(t.words[d], bias, a.table[d], bits) : kind[d] == VIEW_FOR16 ? star_test_words<2>(t.words[d], bias, a.table[d], bits)
                                                                                                                   : star_test_words<4>(t.words[d], bias, a.table[d], bits);
  }
  // The first dimension whose bits did not fit LDS, where a word slot is free (N_STREAM): its foreign keys are streamed like the others'
  // -- coalesced, requested a tile ahead -- and only the bits of the rows that are still alive are asked in global memory (the table stays in
  // the L2).  Reading those rows' keys one by one instead cost a 128-byte line per 4-byte key -- at SSB Q4.1's 4 % of surviving rows nearly the
  // column's bytes -- behind two dependent round trips per row (the probe 481 -> 389 us; traffic 2.04 -> 2.22 GB, profiles/r06_star_traffic.txt).
#pragma unroll
  for (uint32_t d = N_LDS; d < N_LDS + N_STREAM; ++d) {
    if (!__any(alive != 0)) break;
    const StarTable& table = a.table[d];
    const uint32_t bias = kind[d] == VIEW_INT32 ? 0u : t.bias[d];
    const uint32_t delta = bias - table.key_min;
    const uint32_t width = kind[d] == VIEW_FOR8 ? 1u : kind[d] == VIEW_FOR16 ? 2u : 4u;
    const uint32_t w[8] = {t.words[d][0].x, t.words[d][0].y, t.words[d][0].z, t.words[d][0].w, t.words[d][1].x, t.words[d][1].y, t.words[d][1].z, t.words[d][1].w};
    typedef __attribute__((address_space(1))) const uint32_t global_word;
    global_word* bits = (global_word*)table.bits;
    const uint32_t beyond = table.range + 1;   // (a clear bit behind the table's last, as in LDS)
    // (asked row by row, under the condition that the row is alive: asking a lane's first two survivors at once without a condition -- or all
    //  eight rows -- was measured slower, 480 / 433 against 390 us: every lane then issues its loads and the pass is bound by their number)
#define HY_STAR_STREAMED_KEY(J, REL)                                                                                                    \
  {                                                                                                                                      \
    const uint32_t byte = (J) * width, i = byte >> 2;                                                                                    \
    const uint32_t a0 = i & 1 ? w[1] : w[0], a1 = i & 1 ? w[3] : w[2], a2 = i & 1 ? w[5] : w[4], a3 = i & 1 ? w[7] : w[6];               \
    const uint32_t b0 = i & 2 ? a1 : a0, b1 = i & 2 ? a3 : a2; /* (w[i] without an indexed register file) */                             \
    const uint32_t dword = i & 4 ? b1 : b0;                                                                                              \
    const uint32_t distance = (width == 4 ? dword : (dword >> ((byte & 3u) * 8)) & (width == 1 ? 0xFFu : 0xFFFFu)) + delta;              \
    REL = distance < beyond ? distance : beyond;                                                                                         \
  }
    uint32_t pending = alive;
    while (pending) {
      const uint32_t j = __ffs(pending) - 1;
      pending &= pending - 1;
      uint32_t rel;
      HY_STAR_STREAMED_KEY(j, rel)
      if (!__builtin_amdgcn_ubfe(bits[rel >> 5], rel, 1)) alive &= ~(1u << j);
    }
#undef HY_STAR_STREAMED_KEY
  }
  // the other dimensions whose bits did not fit: keys and bits asked in global memory, the rows that are still alive only
  for (uint32_t d = N_LDS + N_STREAM; d < a.n_tables; ++d) {
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
  if (lane == 0 && survivors) atomicAdd(&a.counts[tile], survivors);
}

// Persistent workgroups, a tile's words requested while the tile before is looked up (two fixed sets of registers, the loop unrolled by two:
// rotating one set into the other would wait for the loads just issued).
template <uint32_t N_LDS, uint32_t N_STREAM>
__global__ __launch_bounds__(STAR_THREADS) void star_probe_mask(StarArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_star_bits[];
  const uint32_t tid = threadIdx.x;
  for (uint32_t d = 0; d < N_LDS; ++d) {
    const StarTable& table = a.table[d];
    for (uint32_t i = tid; i < table.words; i += STAR_THREADS) s_star_bits[table.lds_word + i] = table.bits[i];
  }
  __syncthreads();   // (the bits are staged: the only barrier)
  const uint32_t first = tid * 8;
  const uint32_t last_tile = a.n_tiles - 1;   // (a tile past the last: the last one's words once more, not used)
  StarTileWords even, odd;
  uint32_t kind_even[STAR_LDS_DIMENSIONS], kind_odd[STAR_LDS_DIMENSIONS], rows_even = 0, rows_odd = 0;
  if (blockIdx.x < a.n_tiles) star_request_tile<N_LDS + N_STREAM>(a, blockIdx.x, first, even, kind_even, &rows_even);
  for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += 2 * gridDim.x) {
    const uint32_t next = tile + gridDim.x, after = tile + 2 * gridDim.x;
    star_request_tile<N_LDS + N_STREAM>(a, next < a.n_tiles ? next : last_tile, first, odd, kind_odd, &rows_odd);
    star_probe_tile<N_LDS, N_STREAM>(a, tile, tid, first, even, kind_even, rows_even, s_star_bits);
    if (next >= a.n_tiles) break;
    star_request_tile<N_LDS + N_STREAM>(a, after < a.n_tiles ? after : last_tile, first, even, kind_even, &rows_even);
    star_probe_tile<N_LDS, N_STREAM>(a, next, tid, first, odd, kind_odd, rows_odd, s_star_bits);
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

// ---- the finish inside the join -----------------------------------------------------------------------------------------------------
// What it replaces: star_emit_rows, one hy_column_export per column the aggregate reads, hy_projection_arithmetic for the aggregates'
// expressions (operators/projection.cpp) and hy_aggregate_hash over the result (aggregate_hash.cpp:317-403, 605-655) -- for plans whose GROUP
// BY columns and aggregate inputs are int / long columns (any encoding column_value reads) of the fact table or of a dimension.  Survivors
// are 0.8 - 1.6 % of an SSB fact table: their cells are gathered once, by the workgroup that holds their mask, and meet in a hash table in
// LDS that the (persistent) workgroup merges into a small global table when it has seen its last tile.
// The groups carry sums / extremes / one count (no NULL input is accepted: a NULL cell raises FLAG_REFUSED and the caller takes the
// RowID path) and their first and last survivor as (tile << 13 | row in tile) -- monotone in the fact table's row order, i.e. in the order
// of the join result's rows; star_finish_compact turns them into the survivors' ranks = the row numbers hy_aggregate_hash would have seen.
constexpr uint32_t STAR_FINISH_KEYS = 4, STAR_FINISH_AGGREGATES = HY_MAX_STAR_AGGREGATES;
#ifndef HY_STAR_FINISH_THREADS
#define HY_STAR_FINISH_THREADS 512
#endif
constexpr uint32_t STAR_FINISH_THREADS = HY_STAR_FINISH_THREADS;   // two workgroups per CU (256 threads x 4: 189 / 319 us against 135 / 244)
constexpr uint32_t STAR_FINISH_LIST = 2048;            // survivors of a group of tiles looked at per pass
constexpr uint32_t STAR_FINISH_SLOTS = 512;            // a workgroup's table in LDS
constexpr uint32_t STAR_FINISH_GLOBAL_SLOTS = 1u << 14;
constexpr uint32_t STAR_FINISH_MAX_GROUPS = 4096;      // what the compacted result holds (more: the RowID path)
constexpr uint32_t STAR_FACT_TABLE = 0xFFFFFFFFu;
enum : uint32_t { STAR_FLAG_REFUSED = 0, STAR_FLAG_GROUPS = 1 };

constexpr uint32_t STAR_FINISH_FACT_COLUMNS = 4;   // distinct fact-table columns the aggregate reads (their slices' views are staged per tile)
constexpr uint32_t STAR_FINISH_ATTRIBUTES = 4;     // distinct dimension columns it reads (one attribute table each)
enum : uint32_t { STAR_COLUMN_NONE = 0, STAR_COLUMN_FACT = 1, STAR_COLUMN_DIMENSION = 2 };

// A column of the join result as star_finish reads it.  FACT: the fact table's column `index` of StarFinishArgs::fact (its slice's view, or cell by
// cell where the view says VIEW_GENERIC).  DIMENSION: attribute table `index` -- the column's value per KEY of its dimension (star_dim_attributes),
// asked with the fact row's foreign key to the table in slot `table`: one load behind the key instead of RowID -> descriptor -> value id -> value.
struct StarDirectPlan;
struct StarFinishColumn {
  uint32_t kind;
  uint32_t index;
  uint32_t table;
  uint32_t data_type;           // HY_TYPE_INT / HY_TYPE_LONG
};
struct StarFinishAggregate {
  StarFinishColumn left, right;
  uint32_t function, op;        // HY_AGG_*; HY_STAR_NO_OP or HY_ARITH_*
  uint32_t type;                // type of the value that is aggregated (expression_common_type of the operands)
  uint32_t reserved;
};
struct StarFinishArgs {
  StarFinishColumn groupby[STAR_FINISH_KEYS];
  StarFinishAggregate aggregates[STAR_FINISH_AGGREGATES];
  uint32_t n_groupby, n_aggregates;
  uint32_t n_fact, n_attributes;
  // the fact table's columns the aggregate reads: fact_slot[c] = c for c < n_fact, else 0 (a load that repeats column 0's)
  const SliceView* fact_views[STAR_FINISH_FACT_COLUMNS];
  const DevSegment* fact_segments[STAR_FINISH_FACT_COLUMNS];
  uint32_t fact_slot[STAR_FINISH_FACT_COLUMNS];
  uint32_t fact_generic;        // bit c: some slices of column c have no view (VIEW_GENERIC): read cell by cell
  // the dimensions' columns it reads, as attribute tables [range + 1] of int64; entries past n_attributes repeat entry 0
  const int64_t* attributes[STAR_FINISH_ATTRIBUTES];
  uint32_t attribute_slot[STAR_FINISH_ATTRIBUTES];     // slot in StarArgs::table of the dimension whose key indexes the table
  uint32_t attribute_key_min[STAR_FINISH_ATTRIBUTES];
  uint32_t debug;               // timing experiments (HY_DEBUG_SWITCHES builds, HY_STAR_DEBUG): 1 no survivor is looked at, 2 no table work, 4 no cells loaded
  uint32_t attribute_mask[STAR_FINISH_ATTRIBUTES];     // all ones; 0 where the plan reads no dimension column at all (entry 0 of a stand-in table)
  uint32_t capacity;            // global table (power of two)
  uint32_t* tags;
  uint64_t* keys;               // [capacity][STAR_FINISH_KEYS]
  uint64_t* values;             // [capacity][STAR_FINISH_AGGREGATES]
  uint32_t* counts;             // [capacity]
  uint32_t* first;              // [capacity] tile << 13 | row
  uint32_t* last;
  uint32_t* flags;              // STAR_FLAG_*
  const long long* extent;      // star_dim_attributes' [attribute][2]
  struct StarDirectPlan* direct;   // star_finish_plan's answer
};
struct StarFinishHeader {       // what the host reads after the plan's last kernel (pinned memory)
  uint32_t refused, n_groups, key_twice, reserved;
  uint64_t total;               // rows of the join result
};

// The attribute tables: per dimension column the aggregate reads, its value at every key of the dimension's rows (blockIdx.y = attribute).
// A NULL cell raises *null_met (the finish carries no NULLs: the caller takes the RowID path).
struct StarAttributeJobs {
  const uint32_t* bits[STAR_FINISH_ATTRIBUTES];   // the dimension's direct table (star_dim_fill): presence bit and packed RowID per key
  const uint32_t* ids[STAR_FINISH_ATTRIBUTES];
  uint64_t keys[STAR_FINISH_ATTRIBUTES];          // range + 1
  const DevSegment* segments[STAR_FINISH_ATTRIBUTES];
  int64_t* out[STAR_FINISH_ATTRIBUTES];
  long long* extent;            // [STAR_FINISH_ATTRIBUTES][2] smallest / largest value of every attribute (star_finish_plan: the direct-mapped groups)
};
// A thread per key VALUE of the dimension: where the key is present, the column's cell at the row the table names.
__global__ __launch_bounds__(256) void star_dim_attributes(StarAttributeJobs jobs, uint32_t* null_met) {
  const uint32_t k = blockIdx.y;
  long long low = INT64_MAX, high = INT64_MIN;
  for (uint64_t rel = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; rel < jobs.keys[k]; rel += static_cast<uint64_t>(gridDim.x) * 256) {
    if (!((jobs.bits[k][rel >> 5] >> (rel & 31)) & 1)) continue;
    const uint32_t id = jobs.ids[k][rel];
    const Value v = column_value(jobs.segments[k], id >> 16, id & 0xFFFFu);
    if (v.is_null) { *null_met = 1; continue; }
    jobs.out[k][rel] = v.i;
    low = v.i < low ? v.i : low;
    high = v.i > high ? v.i : high;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const long long other_low = __shfl_xor(low, d, 64), other_high = __shfl_xor(high, d, 64);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
  }
  __shared__ long long s_low[4], s_high[4];
  if ((threadIdx.x & 63) == 0) { s_low[threadIdx.x >> 6] = low; s_high[threadIdx.x >> 6] = high; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < 4; ++w) { low = s_low[w] < low ? s_low[w] : low; high = s_high[w] > high ? s_high[w] : high; }
    // (an atomic only where the workgroup's values widen what is there: a dimension's attribute takes a handful of values, and thousands of
    //  workgroups on two words took longer than the table)
    if (low <= high) {
      if (low < __hip_atomic_load(jobs.extent + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(jobs.extent + 2 * k, low);
      if (high > __hip_atomic_load(jobs.extent + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(jobs.extent + 2 * k + 1, high);
    }
  }
}

// Row `row` of the chunk a view describes, without a branch on the view's kind and without a conditional load (the loads of a survivor's
// columns are all requested before the first is looked at): the aligned word that holds the stored value, shifted and masked, and the block
// minimum -- for plain int32 values the word at the front of the data, not used.  A VIEW_GENERIC slice reads its first word (not used either).
__device__ __forceinline__ int32_t star_view_value(const SliceView& view, uint32_t row) {
  const uint32_t width = view.kind == VIEW_FOR8 ? 1u : view.kind == VIEW_FOR16 ? 2u : 4u;
  const uint32_t byte = view.kind == VIEW_GENERIC ? 0u : row * width;
  const bool biased = view.kind != VIEW_INT32 && view.kind != VIEW_GENERIC;
  // (pointers out of a descriptor are generic to the compiler: flat loads, which count as LDS traffic too -- every wait for the next view's words
  //  would wait for them; global-address-space loads do not)
  typedef __attribute__((address_space(1))) const uint32_t global_word;
  global_word* minima = (global_word*)(biased ? view.aux : view.data);
  global_word* words = (global_word*)(static_cast<const char*>(view.data) + (byte & ~3u));
  const uint32_t word = *words;
  const uint32_t minimum = minima[biased ? row / HY_FOR_BLOCK_SIZE : 0u];
  const uint32_t stored = width == 4 ? word : (word >> ((byte & 3u) * 8)) & (width == 1 ? 0xFFu : 0xFFFFu);
  return static_cast<int32_t>(stored + (biased ? minimum : 0u));
}

__device__ __forceinline__ uint64_t star_initial_value(uint32_t function) {
  return function == HY_AGG_MIN ? static_cast<uint64_t>(INT64_MAX) : function == HY_AGG_MAX ? static_cast<uint64_t>(INT64_MIN) : 0ull;
}

// Direct-mapped groups: where every GROUP BY column is a dimension's attribute and the product of the attributes' value ranges (of the rows
// that pass the dimensions' filters: star_dim_attributes) is at most STAR_FINISH_SLOTS, a group's place in the tables IS the mixed-radix code of
// its values -- no hash, no key comparison, no slot to claim (SSB Q2.1: 7 years x 40 brands, Q4.1: 7 years x 5 nations).  Decided on the
// device (one workgroup, between the attribute tables and star_finish; nothing comes to the host): the answer, and the global table's
// slots 0 .. n_codes - 1 initialised with their keys.
struct StarDirectPlan {
  uint32_t direct, n_codes;
  uint32_t stride[STAR_FINISH_KEYS], range[STAR_FINISH_KEYS];
  long long low[STAR_FINISH_KEYS];
};
__global__ __launch_bounds__(256) void star_finish_plan(StarFinishArgs f) {
  __shared__ StarDirectPlan s_plan;
  if (threadIdx.x == 0) {
    StarDirectPlan p;
    p.direct = 1;
    uint64_t codes = 1;
    for (uint32_t g = 0; g < STAR_FINISH_KEYS; ++g) {
      p.stride[g] = 0; p.range[g] = 1; p.low[g] = 0;
      if (g >= f.n_groupby) continue;
      if (f.groupby[g].kind != STAR_COLUMN_DIMENSION) { p.direct = 0; continue; }
      const long long low = f.extent[2 * f.groupby[g].index], high = f.extent[2 * f.groupby[g].index + 1];
      if (low > high || static_cast<unsigned long long>(high - low) >= STAR_FINISH_SLOTS) { p.direct = 0; continue; }
      p.low[g] = low;
      p.range[g] = static_cast<uint32_t>(high - low) + 1;
      p.stride[g] = static_cast<uint32_t>(codes);
      codes *= p.range[g];
      if (codes > STAR_FINISH_SLOTS) { p.direct = 0; codes = 1; }
    }
    p.n_codes = p.direct ? static_cast<uint32_t>(codes) : 0u;
    s_plan = p;
    *f.direct = p;
  }
  __syncthreads();
  for (uint32_t code = threadIdx.x; code < s_plan.n_codes; code += 256) {
    for (uint32_t g = 0; g < STAR_FINISH_KEYS; ++g) f.keys[size_t{code} * STAR_FINISH_KEYS + g] = g < f.n_groupby ? static_cast<uint64_t>(s_plan.low[g] + (code / s_plan.stride[g]) % s_plan.range[g]) : 0ull;
    for (uint32_t g = 0; g < STAR_FINISH_AGGREGATES; ++g) f.values[size_t{code} * STAR_FINISH_AGGREGATES + g] = g < f.n_aggregates ? star_initial_value(f.aggregates[g].function) : 0ull;
    f.counts[code] = 0;
    f.first[code] = 0xFFFFFFFFu;
    f.last[code] = 0;   // (the tags: zeroed with the flags)
  }
}

__device__ __forceinline__ uint32_t star_tuple_hash(const uint64_t (&tuple)[STAR_FINISH_KEYS], uint32_t words) {
  uint32_t h = 0x9E3779B9u;
#pragma unroll
  for (uint32_t w = 0; w < STAR_FINISH_KEYS; ++w) {
    if (w >= words) continue;
    h = (h ^ static_cast<uint32_t>(tuple[w])) * 0x85EBCA77u;
    h = (h ^ (h >> 15) ^ static_cast<uint32_t>(tuple[w] >> 32)) * 0xC2B2AE3Du;
  }
  return h ^ (h >> 13);
}

// Find or insert in the workgroup's LDS table; 0xFFFFFFFF: no place within the probe limit.  One retry loop whose every iteration either
// completes claim + initialise + publish or takes no blocking step (lanes of a wave run in lockstep: a lane must never spin on a slot a
// masked-off neighbour holds) -- the discipline of aggregate.hip's tables.
__device__ __forceinline__ uint32_t star_lds_slot(uint32_t* s_tags, uint64_t* s_keys, uint64_t* s_values, uint32_t* s_counts, uint32_t* s_first, uint32_t* s_last, const StarFinishArgs& f,
                                                 const uint64_t (&tuple)[STAR_FINISH_KEYS], uint32_t hash) {
  const uint32_t ready = 0x80000000u | (hash >> 1);
  uint32_t slot = hash & (STAR_FINISH_SLOTS - 1), probes = 0, result = 0xFFFFFFFFu;
  bool done = false;
  while (!done) {
    const uint32_t tag = __hip_atomic_load(&s_tags[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tag == 0) {
      uint32_t expected = 0;
      if (__hip_atomic_compare_exchange_strong(&s_tags[slot], &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
        for (uint32_t w = 0; w < STAR_FINISH_KEYS; ++w) s_keys[slot * STAR_FINISH_KEYS + w] = tuple[w];
        for (uint32_t g = 0; g < f.n_aggregates; ++g) s_values[slot * f.n_aggregates + g] = star_initial_value(f.aggregates[g].function);
        s_counts[slot] = 0;
        s_first[slot] = 0xFFFFFFFFu;
        s_last[slot] = 0;
        __hip_atomic_store(&s_tags[slot], ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        result = slot;
        done = true;
      }
    } else if (tag != 1u) {
      bool equal = tag == ready;
#pragma unroll
      for (uint32_t w = 0; w < STAR_FINISH_KEYS; ++w) equal = equal && (w >= f.n_groupby || s_keys[slot * STAR_FINISH_KEYS + w] == tuple[w]);
      if (equal) { result = slot; done = true; }
      else {
        slot = (slot + 1) & (STAR_FINISH_SLOTS - 1);
        if (++probes >= 64) done = true;
      }
    } else __builtin_amdgcn_s_sleep(1);
  }
  return result;
}

// The same in the global table (agent scope); 0xFFFFFFFF: full.
__device__ __forceinline__ uint32_t star_global_slot(const StarFinishArgs& f, const uint64_t (&tuple)[STAR_FINISH_KEYS], uint32_t hash) {
  const uint32_t ready = 0x80000000u | (hash >> 1);
  uint32_t slot = hash & (f.capacity - 1), probes = 0, result = 0xFFFFFFFFu;
  bool done = false;
  while (!done) {
    const uint32_t tag = __hip_atomic_load(&f.tags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_ACQUIRE);
    if (tag == 0) {
      uint32_t expected = 0;
      if (__hip_atomic_compare_exchange_strong(&f.tags[slot], &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        for (uint32_t w = 0; w < STAR_FINISH_KEYS; ++w) __hip_atomic_store(&f.keys[size_t{slot} * STAR_FINISH_KEYS + w], tuple[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t g = 0; g < STAR_FINISH_AGGREGATES; ++g) {
          __hip_atomic_store(&f.values[size_t{slot} * STAR_FINISH_AGGREGATES + g], g < f.n_aggregates ? star_initial_value(f.aggregates[g].function) : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&f.counts[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&f.first[slot], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&f.last[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&f.tags[slot], ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        result = slot;
        done = true;
      }
    } else if (tag != 1u) {
      bool equal = tag == ready;
      if (equal) {
        for (uint32_t w = 0; w < f.n_groupby; ++w) equal = equal && __hip_atomic_load(&f.keys[size_t{slot} * STAR_FINISH_KEYS + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tuple[w];
      }
      if (equal) { result = slot; done = true; }
      else {
        slot = (slot + 1) & (f.capacity - 1);
        if (++probes >= 512) done = true;
      }
    } else __builtin_amdgcn_s_sleep(1);
  }
  return result;
}

// (a slot's accumulators lie n_aggregates words apart: up to four aggregates leave room for two workgroups per CU)
constexpr size_t star_finish_lds_bytes(uint32_t n_aggregates) {
  return 4 * size_t{STAR_FINISH_LIST} + sizeof(SliceView) * STAR_EMIT_TILES * (HY_MAX_STAR_DIMENSIONS + STAR_FINISH_FACT_COLUMNS) +
         8 * size_t{STAR_FINISH_SLOTS} * (STAR_FINISH_KEYS + (n_aggregates ? n_aggregates : 1)) + 4 * size_t{STAR_FINISH_SLOTS} * 4 + 64;
}

__global__ __launch_bounds__(STAR_FINISH_THREADS, 4) void star_finish(StarArgs a, StarFinishArgs f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_star_finish[];
  uint64_t* s_keys = reinterpret_cast<uint64_t*>(s_star_finish);
  SliceView* s_views = reinterpret_cast<SliceView*>(s_keys + size_t{STAR_FINISH_SLOTS} * STAR_FINISH_KEYS);   // [tile][dimension]: the foreign keys' slices
  SliceView* s_fact_views = s_views + STAR_EMIT_TILES * HY_MAX_STAR_DIMENSIONS;                                 // [tile][fact column]
  uint32_t* s_list = reinterpret_cast<uint32_t*>(s_fact_views + STAR_EMIT_TILES * STAR_FINISH_FACT_COLUMNS);
  uint32_t* s_tags = s_list + STAR_FINISH_LIST;
  uint32_t* s_counts = s_tags + STAR_FINISH_SLOTS;
  uint32_t* s_first = s_counts + STAR_FINISH_SLOTS;
  uint32_t* s_last = s_first + STAR_FINISH_SLOTS;
  uint32_t* s_wave = s_last + STAR_FINISH_SLOTS;   // [waves] + [1] the workgroup gave up
  uint64_t* s_values = reinterpret_cast<uint64_t*>(s_wave + 16);   // [STAR_FINISH_SLOTS][n_aggregates]
  constexpr uint32_t WAVES = STAR_FINISH_THREADS / 64, WORDS = STAR_EMIT_TILES * SLICE_ROWS / 32 / STAR_FINISH_THREADS;   // mask words per thread: 8
  static_assert(WORDS % 4 == 0 && STAR_EMIT_TILES * (HY_MAX_STAR_DIMENSIONS + STAR_FINISH_FACT_COLUMNS) <= STAR_FINISH_THREADS, "a thread's mask words are 16-byte pieces; one view per thread");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // direct-mapped groups (star_finish_plan) or the hash table
  const uint32_t n_codes = f.direct->n_codes;
  const bool direct = f.direct->direct != 0;
  uint32_t stride[STAR_FINISH_KEYS];
  long long low[STAR_FINISH_KEYS];
#pragma unroll
  for (uint32_t g = 0; g < STAR_FINISH_KEYS; ++g) { stride[g] = f.direct->stride[g]; low[g] = f.direct->low[g]; }
  for (uint32_t i = tid; i < STAR_FINISH_SLOTS; i += STAR_FINISH_THREADS) {
    s_tags[i] = 0;
    if (direct && i < n_codes) {
      s_counts[i] = 0;
      s_first[i] = 0xFFFFFFFFu;
      s_last[i] = 0;
      for (uint32_t g = 0; g < f.n_aggregates; ++g) s_values[i * f.n_aggregates + g] = star_initial_value(f.aggregates[g].function);
    }
  }
  if (tid == 0) s_wave[WAVES] = 0;
  const uint32_t n_groups_of_tiles = (a.n_tiles + STAR_EMIT_TILES - 1) / STAR_EMIT_TILES;
  // A group of tiles costs a round trip for its masks and views before the first survivor is known: the NEXT group's are requested while this
  // one's survivors are worked on.  Thread t < 64 stages the foreign-key view (tile t / 8, dimension t % 8), thread 64 <= t < 96 a fact column's.
  const bool stages_key = tid < STAR_EMIT_TILES * HY_MAX_STAR_DIMENSIONS && (tid % HY_MAX_STAR_DIMENSIONS) < a.n_tables;
  const uint32_t fact_index = tid - STAR_EMIT_TILES * HY_MAX_STAR_DIMENSIONS;
  const bool stages_fact = fact_index < STAR_EMIT_TILES * STAR_FINISH_FACT_COLUMNS && (fact_index % STAR_FINISH_FACT_COLUMNS) < f.n_fact;
  const uint32_t my_tile = stages_key ? tid / HY_MAX_STAR_DIMENSIONS : stages_fact ? fact_index / STAR_FINISH_FACT_COLUMNS : 0u;
  const SliceView* my_views = a.table[0].views;   // (every thread loads a view -- the others the first tile's first, nobody's: no load under a condition)
#pragma unroll
  for (uint32_t d = 0; d < HY_MAX_STAR_DIMENSIONS; ++d) my_views = stages_key && tid % HY_MAX_STAR_DIMENSIONS == d ? a.table[d].views : my_views;
#pragma unroll
  for (uint32_t c = 0; c < STAR_FINISH_FACT_COLUMNS; ++c) my_views = stages_fact && fact_index % STAR_FINISH_FACT_COLUMNS == c ? f.fact_views[c] : my_views;
  SliceView* my_place = stages_key ? s_views + tid : stages_fact ? s_fact_views + fact_index : nullptr;
  u32x4_t next_mask[WORDS / 4];
  const void* next_data = nullptr;     // (the view's six words one by one: a struct the compiler may not keep in registers)
  const void* next_aux = nullptr;
  uint32_t next_chunk = 0, next_row_begin = 0, next_row_count = 0, next_kind = 0;
#define HY_STAR_REQUEST(GROUP)                                                                                                       \
  {   /* (a group past the last: the first tile's, not used) */                                                                      \
    const uint32_t r_first = (GROUP) * STAR_EMIT_TILES < a.n_tiles ? (GROUP) * STAR_EMIT_TILES : 0u;                                 \
    const uint32_t r_tiles = a.n_tiles - r_first < STAR_EMIT_TILES ? a.n_tiles - r_first : STAR_EMIT_TILES;                          \
    const u32x4_t* r_pieces = reinterpret_cast<const u32x4_t*>(a.masks + static_cast<size_t>(r_first) * STAR_THREADS);               \
    _Pragma("unroll") for (uint32_t k = 0; k < WORDS / 4; ++k) {                                                                     \
      const uint32_t piece = tid * (WORDS / 4) + k;                                                                                  \
      next_mask[k] = r_pieces[piece < r_tiles * (STAR_THREADS / 16) ? piece : 0u];                                                   \
    }                                                                                                                                \
    const SliceView* r_view = my_views + r_first + (my_tile < r_tiles ? my_tile : 0u);                                               \
    next_data = r_view->data;                                                                                                        \
    next_aux = r_view->aux;                                                                                                          \
    next_chunk = r_view->chunk;                                                                                                      \
    next_row_begin = r_view->row_begin;                                                                                              \
    next_row_count = r_view->row_count;                                                                                              \
    next_kind = r_view->kind;                                                                                                        \
  }
  HY_STAR_REQUEST(blockIdx.x)
  for (uint32_t group = blockIdx.x; group < n_groups_of_tiles; group += gridDim.x) {
    const uint32_t first_tile = group * STAR_EMIT_TILES;
    const uint32_t n_tiles = a.n_tiles - first_tile < STAR_EMIT_TILES ? a.n_tiles - first_tile : STAR_EMIT_TILES;
    __syncthreads();   // (the list and the views of the group before are done with; the tags are cleared)
    if (my_place) { my_place->data = next_data; my_place->aux = next_aux; my_place->chunk = next_chunk; my_place->row_begin = next_row_begin; my_place->row_count = next_row_count; my_place->kind = next_kind; }
    uint32_t mask[WORDS], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < WORDS; ++k) {
      const uint32_t piece = tid * (WORDS / 4) + k / 4;
      const u32x4_t words = next_mask[k / 4];
      const uint32_t word = k % 4 == 0 ? words.x : k % 4 == 1 ? words.y : k % 4 == 2 ? words.z : words.w;
      mask[k] = piece < n_tiles * (STAR_THREADS / 16) ? word : 0u;
      mine += __popc(mask[k]);
    }
    HY_STAR_REQUEST(group + gridDim.x)
    uint32_t inclusive = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t other = __shfl_up(inclusive, d, 64);
      if (lane >= static_cast<uint32_t>(d)) inclusive += other;
    }
    if (lane == 63) s_wave[wave] = inclusive;
    __syncthreads();
    uint32_t before = inclusive - mine, total = 0;
    for (uint32_t w = 0; w < WAVES; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
    for (uint32_t pass_begin = 0; pass_begin < total; pass_begin += STAR_FINISH_LIST) {
      if (pass_begin) __syncthreads();
      uint32_t rank = before;
#pragma unroll
      for (uint32_t k = 0; k < WORDS; ++k) {
        uint32_t bits = mask[k];
        const uint32_t row0 = (tid * WORDS + k) * 32;
        while (bits) {
          const uint32_t j = __ffs(bits) - 1;
          bits &= bits - 1;
          if (rank - pass_begin < STAR_FINISH_LIST) s_list[rank - pass_begin] = row0 + j;
          ++rank;
        }
      }
      __syncthreads();
      const uint32_t in_pass = total - pass_begin < STAR_FINISH_LIST ? total - pass_begin : STAR_FINISH_LIST;
      for (uint32_t i = tid; i < in_pass; i += STAR_FINISH_THREADS) {
        if (f.debug & 1) break;
        const uint32_t entry = s_list[i];
        const uint32_t tile = entry >> 13, row = f.debug & 4 ? 0u : entry & 8191u;
        const SliceView* views = s_views + (f.debug & 4 ? 0u : tile) * HY_MAX_STAR_DIMENSIONS;
        const SliceView* fact_views = s_fact_views + tile * STAR_FINISH_FACT_COLUMNS;
        // round trip 1: the row's foreign keys to the dimensions somebody reads, and its cells of the fact table's columns.  Every load below is
        // unconditional (entries past the plan's repeat entry 0: the same address once more) -- a load on some control-flow paths only makes the
        // compiler wait for everything in flight where the paths meet, and the row's dozen loads would go one round trip at a time.
        uint32_t rel[STAR_FINISH_ATTRIBUTES];
#pragma unroll
        for (uint32_t k = 0; k < STAR_FINISH_ATTRIBUTES; ++k) {
          const SliceView& view = views[f.attribute_slot[k]];
          rel[k] = (static_cast<uint32_t>(star_view_value(view, view.row_begin + row)) - f.attribute_key_min[k]) & f.attribute_mask[k];
        }
        int64_t fact[STAR_FINISH_FACT_COLUMNS];
#pragma unroll
        for (uint32_t c = 0; c < STAR_FINISH_FACT_COLUMNS; ++c) {
          const SliceView& view = fact_views[f.fact_slot[c]];
          fact[c] = star_view_value(view, view.row_begin + row);
        }
        // round trip 2: the dimensions' attributes at those keys
        int64_t attribute[STAR_FINISH_ATTRIBUTES];
#pragma unroll
        for (uint32_t k = 0; k < STAR_FINISH_ATTRIBUTES; ++k) attribute[k] = f.attributes[k][rel[k]];
        bool refuse = false;
        if (f.fact_generic) {   // columns with slices no view describes (dictionary segments, nullable ones, int64): those rows cell by cell
#pragma unroll
          for (uint32_t c = 0; c < STAR_FINISH_FACT_COLUMNS; ++c) {
            if (!((f.fact_generic >> c) & 1)) continue;
            const SliceView& view = fact_views[c];
            if (view.kind != VIEW_GENERIC) continue;
            const Value v = column_value(f.fact_segments[c], view.chunk, view.row_begin + row);
            refuse = refuse || v.is_null;
            fact[c] = v.i;
          }
        }
        auto cell = [&](const StarFinishColumn& column) -> int64_t {
          int64_t v = 0;
          if (column.kind == STAR_COLUMN_FACT) {
#pragma unroll
            for (uint32_t c = 0; c < STAR_FINISH_FACT_COLUMNS; ++c) v = column.index == c ? fact[c] : v;
          } else {
#pragma unroll
            for (uint32_t k = 0; k < STAR_FINISH_ATTRIBUTES; ++k) v = column.index == k ? attribute[k] : v;
          }
          return v;
        };
        uint64_t tuple[STAR_FINISH_KEYS];
#pragma unroll
        for (uint32_t g = 0; g < STAR_FINISH_KEYS; ++g) tuple[g] = g < f.n_groupby ? static_cast<uint64_t>(cell(f.groupby[g])) : 0ull;
        // an aggregate's input at this row; *null_cell: the expression's cell is NULL (division by zero) -- the plan goes to the RowID path
        auto input_of = [&](uint32_t g, bool* null_cell) -> uint64_t {
          const StarFinishAggregate& spec = f.aggregates[g];
          Value v{false, cell(spec.left), 0.0};
          if (spec.op != HY_STAR_NO_OP) {
            const Value right{false, cell(spec.right), 0.0};
            Value out{false, 0, 0.0};
            *null_cell = arithmetic_cell(spec.op, spec.left.data_type, spec.right.data_type, spec.type, v, right, &out) || *null_cell;
            v = out;
          }
          return static_cast<uint64_t>(v.i);
        };
        const uint32_t hash = star_tuple_hash(tuple, f.n_groupby);
        const uint32_t position = first_tile * SLICE_ROWS + entry;   // (tile << 13 | row, tiles counted from the table's first)
        if (f.debug & 2) { if (hash == 0x12345u) s_wave[WAVES] = 1; continue; }
        uint32_t slot = 0xFFFFFFFFu;
        if (direct) {   // the mixed-radix code of the row's values is its group's place
          uint32_t code = 0;
#pragma unroll
          for (uint32_t g = 0; g < STAR_FINISH_KEYS; ++g) code += static_cast<uint32_t>(static_cast<long long>(tuple[g]) - low[g]) * stride[g];
          if (code < n_codes) slot = code;
          else { s_wave[WAVES] = 1; continue; }   // (a value outside the extent its attribute table was built with: cannot happen)
        } else slot = star_lds_slot(s_tags, s_keys, s_values, s_counts, s_first, s_last, f, tuple, hash);
        if (slot != 0xFFFFFFFFu) {
          for (uint32_t g = 0; g < f.n_aggregates; ++g) {
            if (f.aggregates[g].left.kind == STAR_COLUMN_NONE) continue;
            const uint32_t function = f.aggregates[g].function;
            const uint64_t input = input_of(g, &refuse);
            uint64_t* target = &s_values[slot * f.n_aggregates + g];
            if (function == HY_AGG_MIN) atomicMin(reinterpret_cast<long long*>(target), static_cast<long long>(input));
            else if (function == HY_AGG_MAX) atomicMax(reinterpret_cast<long long*>(target), static_cast<long long>(input));
            else if (function != HY_AGG_COUNT) atomicAdd(reinterpret_cast<unsigned long long*>(target), static_cast<unsigned long long>(input));
          }
          atomicAdd(&s_counts[slot], 1u);
          if (position < s_first[slot]) atomicMin(&s_first[slot], position);
          if (position > s_last[slot]) atomicMax(&s_last[slot], position);
        } else {   // the workgroup's table has no place for this group: the row goes to the global table itself
          const uint32_t global = star_global_slot(f, tuple, hash);
          if (global == 0xFFFFFFFFu) { s_wave[WAVES] = 1; continue; }
          for (uint32_t g = 0; g < f.n_aggregates; ++g) {
            if (f.aggregates[g].left.kind == STAR_COLUMN_NONE) continue;
            const uint32_t function = f.aggregates[g].function;
            const uint64_t input = input_of(g, &refuse);
            uint64_t* target = &f.values[size_t{global} * STAR_FINISH_AGGREGATES + g];
            if (function == HY_AGG_MIN) atomicMin(reinterpret_cast<long long*>(target), static_cast<long long>(input));
            else if (function == HY_AGG_MAX) atomicMax(reinterpret_cast<long long*>(target), static_cast<long long>(input));
            else if (function != HY_AGG_COUNT) atomicAdd(reinterpret_cast<unsigned long long*>(target), static_cast<unsigned long long>(input));
          }
          atomicAdd(&f.counts[global], 1u);
          atomicMin(&f.first[global], position);
          atomicMax(&f.last[global], position);
        }
        if (refuse) s_wave[WAVES] = 1;   // (a NULL cell: what this row left in the tables does not matter, the result is discarded)
      }
    }
  }
  __syncthreads();
  // the workgroup's groups -> the global table
  bool gave_up = s_wave[WAVES] != 0;
  for (uint32_t slot = tid; slot < STAR_FINISH_SLOTS; slot += STAR_FINISH_THREADS) {
    uint32_t global = slot;
    if (direct) {   // the global table is direct-mapped too: its keys are in place (star_finish_plan)
      if (slot >= n_codes || s_counts[slot] == 0) continue;
      f.tags[slot] = 0x80000000u;
    } else {
      if (s_tags[slot] == 0) continue;
      uint64_t tuple[STAR_FINISH_KEYS];
#pragma unroll
      for (uint32_t w = 0; w < STAR_FINISH_KEYS; ++w) tuple[w] = s_keys[slot * STAR_FINISH_KEYS + w];
      global = star_global_slot(f, tuple, star_tuple_hash(tuple, f.n_groupby));
    }
    if (global == 0xFFFFFFFFu) { gave_up = true; continue; }
    for (uint32_t g = 0; g < f.n_aggregates; ++g) {
      if (f.aggregates[g].left.kind == STAR_COLUMN_NONE) continue;
      const uint32_t function = f.aggregates[g].function;
      uint64_t* target = &f.values[size_t{global} * STAR_FINISH_AGGREGATES + g];
      const uint64_t value = s_values[slot * f.n_aggregates + g];
      if (function == HY_AGG_MIN) atomicMin(reinterpret_cast<long long*>(target), static_cast<long long>(value));
      else if (function == HY_AGG_MAX) atomicMax(reinterpret_cast<long long*>(target), static_cast<long long>(value));
      else if (function != HY_AGG_COUNT) atomicAdd(reinterpret_cast<unsigned long long*>(target), static_cast<unsigned long long>(value));
    }
    atomicAdd(&f.counts[global], s_counts[slot]);
    atomicMin(&f.first[global], s_first[slot]);
    atomicMax(&f.last[global], s_last[slot]);
  }
  if (gave_up) f.flags[STAR_FLAG_REFUSED] = 1;
}

#undef HY_STAR_REQUEST

// The global table's groups, densely, into the block the host reads (pinned memory): keys, first / last as RANKS among the survivors (the row
// numbers of the join result: survivors of the tiles before + set mask bits in front of the row), values, counts; the workgroup that finishes
// last writes the header.  One slot per thread; flags[STAR_FLAG_GROUPS] hands out the places (the host orders the groups anyway).
constexpr uint32_t STAR_FLAG_TICKET = 2, STAR_FLAG_NULL_ATTRIBUTE = 3;
__global__ __launch_bounds__(256) void star_finish_compact(StarArgs a, StarFinishArgs f, const uint32_t* duplicate, StarFinishHeader* header, uint64_t* out_keys, uint64_t* out_first,
                                                           uint64_t* out_last, uint64_t* out_values, uint64_t* out_counts) {
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  auto rank_of = [&](uint32_t position) -> uint64_t {
    const uint32_t tile = position >> 13, row = position & 8191u;
    const u32x4_t* quads = reinterpret_cast<const u32x4_t*>(a.masks + static_cast<size_t>(tile) * STAR_THREADS);
    uint32_t bits = 0;
    const uint32_t full = row / 128;   // whole 16-byte pieces in front of the row: up to 63, sixteen loads in flight
    for (uint32_t q = 0; q < full; q += 16) {
      u32x4_t piece[16];
#pragma unroll
      for (uint32_t k = 0; k < 16; ++k) piece[k] = quads[q + k < full ? q + k : q];
#pragma unroll
      for (uint32_t k = 0; k < 16; ++k) if (q + k < full) bits += __popc(piece[k].x) + __popc(piece[k].y) + __popc(piece[k].z) + __popc(piece[k].w);
    }
    const u32x4_t last_piece = quads[full];
    const uint32_t in_piece = row & 127u;
    const uint32_t w[4] = {last_piece.x, last_piece.y, last_piece.z, last_piece.w};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      if (in_piece >= 32 * (k + 1)) bits += __popc(w[k]);
      else if (in_piece > 32 * k) bits += __popc(w[k] & ((1u << (in_piece - 32 * k)) - 1u));
    }
    return a.base[tile] + bits;
  };
  const uint32_t slot = blockIdx.x * 256 + tid;
  const bool taken = slot < f.capacity && f.tags[slot] != 0;
  const uint64_t peers = __ballot(taken);
  uint32_t place = 0;
  if (peers) {
    if (lane == static_cast<uint32_t>(__ffsll(static_cast<long long>(peers)) - 1)) place = atomicAdd(&f.flags[STAR_FLAG_GROUPS], static_cast<uint32_t>(__popcll(peers)));
    place = __shfl(place, __ffsll(static_cast<long long>(peers)) - 1, 64) + static_cast<uint32_t>(__popcll(peers & ((1ull << lane) - 1)));
  }
  if (taken && place < STAR_FINISH_MAX_GROUPS) {
    for (uint32_t w = 0; w < STAR_FINISH_KEYS; ++w) out_keys[size_t{place} * STAR_FINISH_KEYS + w] = f.keys[size_t{slot} * STAR_FINISH_KEYS + w];
    for (uint32_t g = 0; g < STAR_FINISH_AGGREGATES; ++g) out_values[size_t{place} * STAR_FINISH_AGGREGATES + g] = f.values[size_t{slot} * STAR_FINISH_AGGREGATES + g];
    out_counts[place] = f.counts[slot];
    out_first[place] = rank_of(f.first[slot]);
    out_last[place] = rank_of(f.last[slot]);
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0 && atomicAdd(&f.flags[STAR_FLAG_TICKET], 1u) == gridDim.x - 1) {
    const uint32_t n_groups = atomicAdd(&f.flags[STAR_FLAG_GROUPS], 0u);
    header->refused = atomicAdd(&f.flags[STAR_FLAG_REFUSED], 0u) | atomicAdd(&f.flags[STAR_FLAG_NULL_ATTRIBUTE], 0u) | (n_groups > STAR_FINISH_MAX_GROUPS ? 1u : 0u);
    header->n_groups = n_groups;
    header->key_twice = *duplicate;
    header->total = a.base[a.n_tiles];
    __threadfence_system();
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
// Can star_finish read this column cell by cell (column_value), as an integer?
static bool star_finish_reads(const hy_column* column) {
  if (!column || column->is_reference || column->is_mvcc || column->has_compressed || column->has_dictionary_without_values) return false;
  return column->data_type == HY_TYPE_INT || column->data_type == HY_TYPE_LONG;
}

// Is the plan's aggregate one star_finish computes?  (MIN / MAX / SUM / AVG / COUNT over int / long cells, one to four int / long GROUP BY columns)
static bool star_finish_applies(const StarFinishRequest* finish, uint32_t n_dimensions, const hy_column* fact_shape) {
  if (!finish || !option(HY_OPT_STAR_FUSED_FINISH)) return false;
  if (finish->n_groupby == 0 || finish->n_groupby > STAR_FINISH_KEYS || finish->n_aggregates > STAR_FINISH_AGGREGATES) return false;
  if (fact_shape->rows >= (1ull << 31) || fact_shape->n_slices >= (1u << 19)) return false;   // (positions are tile << 13 | row in 32 bits, counts 32 bits)
  auto column_ok = [&](const StarFinishColumnSpec& c) { return c.table <= n_dimensions && star_finish_reads(c.column); };
  for (uint32_t g = 0; g < finish->n_groupby; ++g) if (!column_ok(finish->groupby[g])) return false;
  for (uint32_t g = 0; g < finish->n_aggregates; ++g) {
    const auto& spec = finish->aggregates[g];
    if (spec.function != HY_AGG_MIN && spec.function != HY_AGG_MAX && spec.function != HY_AGG_SUM && spec.function != HY_AGG_AVG && spec.function != HY_AGG_COUNT) return false;
    if (!spec.left.column) { if (spec.function != HY_AGG_COUNT || spec.op != HY_STAR_NO_OP) return false; continue; }
    if (!column_ok(spec.left)) return false;
    if (spec.op != HY_STAR_NO_OP && (spec.op > HY_ARITH_MOD || !column_ok(spec.right))) return false;
  }
  return true;
}

hy_status star_probe_rows(const StarProbeDimension* dimensions, uint32_t n_dimensions, DeviceBuffer& fact_rows, std::vector<std::unique_ptr<DeviceBuffer>>& dimension_rows, uint64_t* n_rows,
                          bool* applicable, const StarFinishRequest* finish, StarFinishGroups* groups) {
  *applicable = false;
  *n_rows = 0;
  if (groups) { groups->done = false; groups->n_groups = 0; }
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
  // ---- the aggregate inside the join (star_finish): which columns it reads ----------------------------------------------------------------
  struct FinishPlan {
    bool on = false;
    std::vector<const hy_column*> fact;                                  // distinct fact-table columns
    std::vector<std::pair<uint32_t, const hy_column*>> attributes;       // distinct (dimension, column)
    std::vector<std::unique_ptr<DeviceBuffer>> attribute_tables;
    DeviceBuffer table;                                                  // flags | tags | counts | first | last | keys | values
    DeviceBuffer extents;                                                // [STAR_FINISH_ATTRIBUTES][2] smallest / largest attribute value | StarDirectPlan
  } plan;
  plan.on = groups && star_finish_applies(finish, n_dimensions, shape);
  auto plan_column = [&](const StarFinishColumnSpec& c) {
    if (!c.column) return;
    if (c.table == 0) { if (std::find(plan.fact.begin(), plan.fact.end(), c.column) == plan.fact.end()) plan.fact.push_back(c.column); }
    else if (std::find(plan.attributes.begin(), plan.attributes.end(), std::make_pair(c.table - 1, c.column)) == plan.attributes.end()) plan.attributes.emplace_back(c.table - 1, c.column);
  };
  if (plan.on) {
    for (uint32_t g = 0; g < finish->n_groupby; ++g) plan_column(finish->groupby[g]);
    for (uint32_t g = 0; g < finish->n_aggregates; ++g) {
      plan_column(finish->aggregates[g].left);
      if (finish->aggregates[g].op != HY_STAR_NO_OP) plan_column(finish->aggregates[g].right);
    }
    plan.on = plan.fact.size() <= STAR_FINISH_FACT_COLUMNS && plan.attributes.size() <= STAR_FINISH_ATTRIBUTES;
    for (const hy_column* c : plan.fact) plan.on = plan.on && c->n_slices == shape->n_slices;
  }
  constexpr size_t FINISH_TAGS_BYTES = 4 * size_t{STAR_FINISH_GLOBAL_SLOTS}, FINISH_KEYS_BYTES = 8 * size_t{STAR_FINISH_GLOBAL_SLOTS} * STAR_FINISH_KEYS;
  constexpr size_t FINISH_VALUES_BYTES = 8 * size_t{STAR_FINISH_GLOBAL_SLOTS} * STAR_FINISH_AGGREGATES;
  if (plan.on) {
    HY_TRY(plan.table.alloc(64 + 4 * FINISH_TAGS_BYTES + FINISH_KEYS_BYTES + FINISH_VALUES_BYTES));
    HY_HIP(hipMemsetAsync(plan.table.ptr, 0, 64 + FINISH_TAGS_BYTES, stream));   // flags and tags
    HY_TRY(plan.extents.alloc(16 * STAR_FINISH_ATTRIBUTES + sizeof(StarDirectPlan) + 64));
    long long nothing[2 * STAR_FINISH_ATTRIBUTES];
    for (uint32_t k = 0; k < STAR_FINISH_ATTRIBUTES; ++k) { nothing[2 * k] = INT64_MAX; nothing[2 * k + 1] = INT64_MIN; }
    HY_HIP(hipMemcpyAsync(plan.extents.ptr, nothing, sizeof(nothing), hipMemcpyHostToDevice, stream));   // (pageable memory: copied before the call returns)
  }
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
    jobs.row_base[d] = dimensions[d].key->d_row_base;
    jobs.n_chunks[d] = dimensions[d].key->n_chunks;
    {   // chunks of one size (all but the last): the row -> RowID arithmetic needs no search
      const hy_column* key = dimensions[d].key;
      uint32_t size = key->n_chunks ? key->host_segments[0].size : 0;
      for (uint32_t c = 0; c + 1 < key->n_chunks && size; ++c) if (key->host_segments[c].size != size) size = 0;
      if (key->n_chunks && key->host_segments[key->n_chunks - 1].size > size) size = 0;
      jobs.chunk_rows[d] = size;
    }
    jobs.filter_segments[d] = dimensions[d].filter ? dimensions[d].filter->d_segments : nullptr;
    jobs.filter_jobs[d] = dimensions[d].filter_jobs;
    most_rows = std::max(most_rows, dimensions[d].n_rows);
  }
  // (a row per thread: a row is three dependent round trips and an atomic that is waited for -- 256 workgroups walking a million rows spent 59 us on it)
  const uint32_t job_grid = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((most_rows + 255) / 256, 8192)));
  // ---- the direct tables -------------------------------------------------------------------------------------------------------
  struct Built { uint32_t* bits = nullptr; DeviceBuffer ids; int64_t key_min = 0; uint64_t range = 0; uint32_t words = 0; bool empty = false; };
  std::vector<std::unique_ptr<Built>> built(n_dimensions);
  DeviceBuffer duplicate;   // [16 words of flags] | every dimension's presence words: ONE buffer, one memset
  bool nothing_joins = false;
  size_t all_words = 16;
  for (uint32_t d = 0; d < n_dimensions; ++d) {
    built[d] = std::make_unique<Built>();
    Built& b = *built[d];
    if (extent[2 * d] > extent[2 * d + 1]) { b.empty = true; nothing_joins = true; continue; }   // no row (or only NULL keys): an Inner join with it is empty
    const int64_t low = static_cast<int64_t>(extent[2 * d] ^ SIGN), high = static_cast<int64_t>(extent[2 * d + 1] ^ SIGN);
    if (low < INT32_MIN || high > INT32_MAX || static_cast<uint64_t>(high - low) > STAR_MAX_RANGE) return HY_OK;   // (a sparse key: the rank table / directory of hy_join_hash)
    b.key_min = low;
    b.range = static_cast<uint64_t>(high - low);
    b.words = static_cast<uint32_t>((b.range + 1) >> 5) + 1;   // (one bit more than the range: what keys outside it ask, always zero)
    all_words += (size_t{b.words} + 7) & ~size_t{7};
  }
  HY_TRY(duplicate.alloc(4 * all_words + 64));
  HY_HIP(hipMemsetAsync(duplicate.ptr, 0, 4 * all_words, stream));
  {
    size_t at = 16;
    for (uint32_t d = 0; d < n_dimensions; ++d) {
      Built& b = *built[d];
      if (b.empty) continue;
      b.bits = duplicate.as<uint32_t>() + at;
      at += (size_t{b.words} + 7) & ~size_t{7};
      HY_TRY(b.ids.alloc(4 * (b.range + 1) + 16));
      jobs.key_min[d] = b.key_min;
      jobs.bits[d] = b.bits;
      jobs.ids[d] = b.ids.as<uint32_t>();
    }
  }
  for (uint32_t d = 0; d < n_dimensions; ++d) if (built[d]->empty) { jobs.n[d] = 0; jobs.n_in_memory[d] = nullptr; }
  if (!nothing_joins) hipLaunchKernelGGL(star_dim_fill, dim3(job_grid, n_dimensions), dim3(256), 0, stream, jobs, duplicate.as<uint32_t>());
  if (!nothing_joins && plan.on && !plan.attributes.empty()) {   // what the aggregate reads of the dimensions, per key
    StarAttributeJobs attribute_jobs;
    std::memset(&attribute_jobs, 0, sizeof(attribute_jobs));
    uint64_t most_keys = 1;
    for (size_t k = 0; k < plan.attributes.size(); ++k) {
      const uint32_t d = plan.attributes[k].first;
      const hy_column* column = plan.attributes[k].second;
      plan.attribute_tables.push_back(std::make_unique<DeviceBuffer>());
      HY_TRY(plan.attribute_tables.back()->alloc(8 * (built[d]->range + 1) + 16));
      attribute_jobs.bits[k] = jobs.bits[d];
      attribute_jobs.ids[k] = jobs.ids[d];
      attribute_jobs.keys[k] = built[d]->range + 1;
      attribute_jobs.segments[k] = column->d_segments;
      most_keys = std::max(most_keys, built[d]->range + 1);
      attribute_jobs.out[k] = plan.attribute_tables.back()->as<int64_t>();
    }
    attribute_jobs.extent = plan.extents.as<long long>();
    const uint64_t attribute_blocks = (most_keys + 255) / 256;
    const dim3 attribute_grid(static_cast<uint32_t>(attribute_blocks < 512 ? attribute_blocks : 512), static_cast<uint32_t>(plan.attributes.size()));   // (keys without a row cost a coalesced word: a few per thread)
    hipLaunchKernelGGL(star_dim_attributes, attribute_grid, dim3(256), 0, stream, attribute_jobs,
                       reinterpret_cast<uint32_t*>(plan.table.ptr) + STAR_FLAG_NULL_ATTRIBUTE);
  }
  dimension_rows.clear();
  dimension_rows.resize(n_dimensions);
  if (nothing_joins) {   // (still "applicable": the join result is empty)
    if (plan.on) {   // ... and so is the aggregate's
      for (uint32_t g = 0; g < finish->n_aggregates; ++g) {
        const auto& spec = finish->aggregates[g];
        groups->input_type[g] = !spec.left.column ? static_cast<uint32_t>(HY_TYPE_LONG) : spec.op == HY_STAR_NO_OP ? spec.left.column->data_type : expression_common_type(spec.left.column->data_type, spec.right.column->data_type);
      }
      groups->keys.clear(); groups->first.clear(); groups->last.clear(); groups->values.clear(); groups->counts.clear();
      groups->done = true;
      *applicable = true;
      return HY_OK;
    }
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
    t.bits = built[d]->bits;
    t.ids = built[d]->ids.as<uint32_t>();
    t.key_min = static_cast<uint32_t>(static_cast<int32_t>(built[d]->key_min));
    t.range = static_cast<uint32_t>(built[d]->range);
    t.words = built[d]->words;
  }
  a.n_tiles = shape->n_slices;
  DeviceBuffer masks, counts, base;
  HY_TRY(masks.alloc(size_t{a.n_tiles} * STAR_THREADS + 16));
  HY_TRY(counts.alloc(4 * (size_t{a.n_tiles} + 1)));
  HY_HIP(hipMemsetAsync(counts.ptr, 0, 4 * (size_t{a.n_tiles} + 1), stream));   // (the probe's waves ADD their survivors)
  HY_TRY(base.alloc(8 * (size_t{a.n_tiles} + 2)));
  a.masks = masks.as<uint8_t>();
  a.counts = counts.as<uint32_t>();
  a.base = base.as<uint64_t>();
  static OncePerDevice lds_raised;
  uint64_t device_bit = 0;
  if (lds_raised.pending(&device_bit)) {
#define HY_STAR_RAISE(L, S) HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(star_probe_mask<L, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAR_LDS_WORDS));
    HY_STAR_RAISE(0, 0) HY_STAR_RAISE(1, 0) HY_STAR_RAISE(2, 0) HY_STAR_RAISE(3, 0) HY_STAR_RAISE(4, 0)
    HY_STAR_RAISE(0, 1) HY_STAR_RAISE(1, 1) HY_STAR_RAISE(2, 1) HY_STAR_RAISE(3, 1)
#undef HY_STAR_RAISE
    lds_raised.done(device_bit);
  }
  const uint32_t probe_grid = std::max(1u, std::min(a.n_tiles, device_cu_count()));
  profile_begin(stream, HY_KERNEL_JOIN_PROBE);
  // (instantiated per number of LDS-resident dimensions, and with or without a streamed one behind them -- the first dimension asked in
  //  global memory, if a word slot is left: the kernel requests exactly their words, unconditionally)
  const uint32_t streamed = a.n_lds < STAR_LDS_DIMENSIONS && a.n_lds < a.n_tables ? 1u : 0u;
#define HY_STAR_LAUNCH(L, S) hipLaunchKernelGGL((star_probe_mask<L, S>), dim3(probe_grid), dim3(STAR_THREADS), 4 * size_t{lds_words}, stream, a)
  switch (a.n_lds * 2 + streamed) {
    case 0: HY_STAR_LAUNCH(0, 0); break;
    case 1: HY_STAR_LAUNCH(0, 1); break;
    case 2: HY_STAR_LAUNCH(1, 0); break;
    case 3: HY_STAR_LAUNCH(1, 1); break;
    case 4: HY_STAR_LAUNCH(2, 0); break;
    case 5: HY_STAR_LAUNCH(2, 1); break;
    case 6: HY_STAR_LAUNCH(3, 0); break;
    case 7: HY_STAR_LAUNCH(3, 1); break;
    default: HY_STAR_LAUNCH(4, 0); break;
  }
#undef HY_STAR_LAUNCH
  profile_end(stream);
  hipLaunchKernelGGL(star_scan_counts, dim3(1), dim3(1024), 0, stream, counts.as<uint32_t>(), base.as<uint64_t>(), a.n_tiles);
  uint64_t total = 0;
  uint32_t key_twice = 0;
  if (plan.on) {
    // ---- the aggregate, where the survivors are (star_finish): one host read for the whole plan -------------------------------------------
    StarFinishArgs f;
    std::memset(&f, 0, sizeof(f));
    f.n_groupby = finish->n_groupby;
    f.n_aggregates = finish->n_aggregates;
    f.n_fact = static_cast<uint32_t>(plan.fact.size());
    f.n_attributes = static_cast<uint32_t>(plan.attributes.size());
    // (entries past the plan's repeat entry 0 -- or, where the plan reads no fact column / no dimension column, the first foreign key and the
    //  finish's own table: loads that go somewhere valid and whose result nobody looks at)
    for (uint32_t c = 0; c < STAR_FINISH_FACT_COLUMNS; ++c) {
      const hy_column* column = plan.fact.empty() ? shape : plan.fact[c < plan.fact.size() ? c : 0];
      f.fact_views[c] = column->d_slice_views;
      f.fact_segments[c] = column->d_segments;
      f.fact_slot[c] = c < plan.fact.size() ? c : 0;
      if (c < plan.fact.size() && !star_reads_fact_key(column)) f.fact_generic |= 1u << c;
    }
    if (plan.fact.empty()) f.n_fact = 1;   // (the views of the first foreign key are staged in its place)
    for (uint32_t k = 0; k < STAR_FINISH_ATTRIBUTES; ++k) {
      if (plan.attributes.empty()) { f.attributes[k] = reinterpret_cast<const int64_t*>(plan.table.ptr); f.attribute_slot[k] = 0; f.attribute_key_min[k] = a.table[0].key_min; f.attribute_mask[k] = 0; continue; }
      const size_t from = k < plan.attributes.size() ? k : 0;
      f.attributes[k] = plan.attribute_tables[from]->as<int64_t>();
      f.attribute_slot[k] = slot_of[plan.attributes[from].first];
      f.attribute_key_min[k] = a.table[f.attribute_slot[k]].key_min;
      f.attribute_mask[k] = 0xFFFFFFFFu;
    }
    auto wire = [&](StarFinishColumn& out, const StarFinishColumnSpec& in) {
      out.kind = STAR_COLUMN_NONE;
      out.data_type = in.column ? in.column->data_type : static_cast<uint32_t>(HY_TYPE_LONG);
      if (!in.column) return;
      if (in.table == 0) {
        out.kind = STAR_COLUMN_FACT;
        out.index = static_cast<uint32_t>(std::find(plan.fact.begin(), plan.fact.end(), in.column) - plan.fact.begin());
      } else {
        out.kind = STAR_COLUMN_DIMENSION;
        out.index = static_cast<uint32_t>(std::find(plan.attributes.begin(), plan.attributes.end(), std::make_pair(in.table - 1, in.column)) - plan.attributes.begin());
        out.table = slot_of[in.table - 1];
      }
    };
    for (uint32_t g = 0; g < f.n_groupby; ++g) wire(f.groupby[g], finish->groupby[g]);
    for (uint32_t g = 0; g < f.n_aggregates; ++g) {
      const auto& spec = finish->aggregates[g];
      StarFinishAggregate& out = f.aggregates[g];
      out.function = spec.function;
      out.op = spec.left.column ? spec.op : HY_STAR_NO_OP;
      wire(out.left, spec.left);
      if (out.op != HY_STAR_NO_OP) wire(out.right, spec.right);
      out.type = !spec.left.column ? static_cast<uint32_t>(HY_TYPE_LONG) : out.op == HY_STAR_NO_OP ? spec.left.column->data_type : expression_common_type(spec.left.column->data_type, spec.right.column->data_type);
      groups->input_type[g] = out.type;
    }
    if (const char* debug = HY_DEBUG_ENV("HY_STAR_DEBUG")) f.debug = static_cast<uint32_t>(atoi(debug));   // timing experiments only
    f.capacity = STAR_FINISH_GLOBAL_SLOTS;
    char* at = plan.table.as<char>();
    f.flags = reinterpret_cast<uint32_t*>(at);                at += 64;
    f.tags = reinterpret_cast<uint32_t*>(at);                 at += FINISH_TAGS_BYTES;
    f.counts = reinterpret_cast<uint32_t*>(at);               at += FINISH_TAGS_BYTES;
    f.first = reinterpret_cast<uint32_t*>(at);                at += FINISH_TAGS_BYTES;
    f.last = reinterpret_cast<uint32_t*>(at);                 at += FINISH_TAGS_BYTES;
    f.keys = reinterpret_cast<uint64_t*>(at);                 at += FINISH_KEYS_BYTES;
    f.values = reinterpret_cast<uint64_t*>(at);
    f.extent = plan.extents.as<long long>();
    f.direct = reinterpret_cast<StarDirectPlan*>(plan.extents.as<char>() + 16 * STAR_FINISH_ATTRIBUTES);
    constexpr size_t G = STAR_FINISH_MAX_GROUPS;
    const size_t staged_bytes = 64 + 8 * G * (STAR_FINISH_KEYS + STAR_FINISH_AGGREGATES + 3);
    unsigned char* staged_host = nullptr;
    unsigned char* staged_dev = nullptr;
    HY_TRY(pinned_staging(staged_bytes, reinterpret_cast<void**>(&staged_host), reinterpret_cast<void**>(&staged_dev)));
    auto arrays = [&](unsigned char* block, uint64_t** keys, uint64_t** first, uint64_t** last, uint64_t** values, uint64_t** group_counts) {
      *keys = reinterpret_cast<uint64_t*>(block + 64);
      *first = *keys + G * STAR_FINISH_KEYS;
      *last = *first + G;
      *values = *last + G;
      *group_counts = *values + G * STAR_FINISH_AGGREGATES;
    };
    static OncePerDevice finish_lds_raised;
    uint64_t finish_bit = 0;
    if (finish_lds_raised.pending(&finish_bit)) {
      HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(star_finish), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(star_finish_lds_bytes(STAR_FINISH_AGGREGATES))));
      finish_lds_raised.done(finish_bit);
    }
    const uint32_t tile_groups = (a.n_tiles + STAR_EMIT_TILES - 1) / STAR_EMIT_TILES;
    hipLaunchKernelGGL(star_finish_plan, dim3(1), dim3(256), 0, stream, f);
    profile_begin(stream, HY_KERNEL_AGGREGATE);
    hipLaunchKernelGGL(star_finish, dim3(std::max(1u, std::min(tile_groups, (1024 / STAR_FINISH_THREADS) * device_cu_count()))), dim3(STAR_FINISH_THREADS), star_finish_lds_bytes(f.n_aggregates), stream, a, f);
    profile_end(stream);
    uint64_t *d_keys, *d_first, *d_last, *d_values, *d_counts;
    arrays(staged_dev, &d_keys, &d_first, &d_last, &d_values, &d_counts);
    hipLaunchKernelGGL(star_finish_compact, dim3(f.capacity / 256), dim3(256), 0, stream, a, f, duplicate.as<uint32_t>(), reinterpret_cast<StarFinishHeader*>(staged_dev), d_keys, d_first, d_last, d_values, d_counts);
    HY_HIP(hipGetLastError());
    HY_HIP(hipStreamSynchronize(stream));
    StarFinishHeader header;
    std::memcpy(&header, staged_host, sizeof(header));
    if (header.key_twice) return HY_OK;   // a dimension key that is not unique: JoinHash's business
    total = header.total;
    if (!header.refused) {
      uint64_t *h_keys, *h_first, *h_last, *h_values, *h_counts;
      arrays(staged_host, &h_keys, &h_first, &h_last, &h_values, &h_counts);
      const size_t n = header.n_groups;
      groups->n_groups = header.n_groups;
      groups->keys.assign(h_keys, h_keys + n * STAR_FINISH_KEYS);
      groups->first.assign(h_first, h_first + n);
      groups->last.assign(h_last, h_last + n);
      groups->values.assign(h_values, h_values + n * STAR_FINISH_AGGREGATES);
      groups->counts.assign(h_counts, h_counts + n);
      groups->done = true;
      dimension_rows.clear();
      dimension_rows.resize(n_dimensions);
      *n_rows = total;
      *applicable = true;
      return HY_OK;
    }
    // (a NULL cell, a division by zero, more groups than the tables hold: the RowIDs after all)
  } else {
    HY_HIP(hipMemcpyAsync(&total, base.as<uint64_t>() + a.n_tiles, 8, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipMemcpyAsync(&key_twice, duplicate.ptr, 4, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
    if (key_twice) return HY_OK;   // a dimension key that is not unique: JoinHash's business
  }
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
