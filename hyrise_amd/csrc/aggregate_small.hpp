// aggregate_small.hpp -- AggregateHash for the TPC-H Q1 shape (included by aggregate.hip, inside namespace hy).
//
// What it replaces (reference, CPU): the row loop of AggregateHash::_aggregate for a handful of groups
// (operators/aggregate_hash.cpp:317-403 get_or_add_result, :605-655 _aggregate_segment, :1016-1176) -- config 4 of BASELINE.json:
// GROUP BY l_returnflag, l_linestatus (dictionary segments, a few distinct values) with SUM / AVG / COUNT over DictionarySegment<float>
// columns.  aggregate_rows handles every encoding x type x function in one 150 KB body at 167 registers and ran this shape at 11 %
// of the HBM roofline; this kernel takes the shape and nothing else:
//   * every GROUP BY column is a dictionary segment in every chunk, the product of their (dictionary size + 1) is at most 16: a
//     row's group is the mixed-radix CODE of its value ids; the first four codes a chunk meets are its DENSE groups (Q1 has four),
//   * every aggregate is SUM / AVG / COUNT over a dictionary-encoded float / double column with 1- or 2-byte value ids (or COUNT(*)),
//   * 1-byte value ids (l_quantity, l_discount): the rows are COUNTED per (dense group, value id) in an LDS histogram -- one LDS
//     atomic per row and column, no dictionary gather at all -- and the counts are weighted with the dictionary once per chunk
//     (the double sums are exact for these columns' products count x value in any order),
//   * 2-byte value ids (l_extendedprice, 240 KB of dictionary per chunk): one gather per row from the chunk's dictionary (one XCD
//     works on one chunk: the dictionary stays in its L2) into four register accumulators selected by the row's dense group.
// One workgroup per chunk, sixteen consecutive rows per lane and step (16-byte loads of 1-byte ids, two for 2-byte ids).  Rows of a
// fifth, sixth ... group of a chunk take LDS atomics on shared cells.  The chunk's groups are merged into the global table like
// aggregate_rows' (global_slot / merge_global): result order, representative rows and values are those of the generic kernel.
// SUM / AVG: double additions in a different order than the reference's row loop -- the stated 1e-9 relative tolerance.
#pragma once

constexpr uint32_t SD_CODES = 16;       // product of (dictionary size + 1) over the GROUP BY columns, at most
constexpr uint32_t SD_DENSE = 4;        // groups of a chunk with register / histogram accumulators
constexpr uint32_t SD_COLUMNS = 4;      // distinct aggregate input columns
constexpr uint32_t SD_NARROW = 2;       // ... of which with 1-byte value ids, at most
constexpr uint32_t SD_WIDE = 2;         // ... and with 2-byte value ids
constexpr uint32_t SD_ROWS = 16;        // consecutive rows of a lane per step
typedef __attribute__((address_space(1))) float global_f32;    // (pointers read from a segment descriptor are generic to the compiler: flat loads, which also count on lgkmcnt)
typedef __attribute__((address_space(1))) double global_f64;

struct SmallDomainPlan {
  uint32_t n_columns, n_narrow;                   // distinct input columns; the first n_narrow have 1-byte value ids, the others 2-byte ones
  const DevSegment* column[SD_COLUMNS];
  uint32_t column_of_aggregate[MAX_AGGREGATES];   // 0xFFFFFFFF: COUNT(*)
  uint32_t key_width[MAX_GROUPBY];                // bytes per value id of the GROUP BY columns (1 or 2; the same in every chunk)
  uint32_t debug;                                 // HY_AGG_SMALL_DEBUG (timing experiments, wrong results): 1 no histograms, 2 no phase 2, 4 phase 2 without the LDS gathers and adds, 8 no dense lookup
};

// value id of row j (0..15) of a lane's sixteen consecutive ids loaded as 16 bytes (WIDTH 1) or 2 x 16 bytes (WIDTH 2)
__device__ __forceinline__ uint32_t sd_id(const u32x4 (&v)[2], uint32_t width, uint32_t j) {
  if (width == 1) {
    const uint32_t w = j < 4 ? v[0].x : j < 8 ? v[0].y : j < 12 ? v[0].z : v[0].w;
    return (w >> (8 * (j & 3))) & 0xFFu;
  }
  const u32x4 half = j < 8 ? v[0] : v[1];
  const uint32_t k = j & 7;
  const uint32_t w = k < 2 ? half.x : k < 4 ? half.y : k < 6 ? half.z : half.w;
  return (w >> (16 * (k & 1))) & 0xFFFFu;
}

// (a load starts at an existing row: it reads less than 16 bytes past the segment's last id -- inside the padding every uploaded buffer has)
// AHEAD: a load issued one step before its use -- volatile, or the compiler sinks it down to that use.
template <bool AHEAD = false>
__device__ __forceinline__ void sd_load_ids(const void* data, uint32_t width, uint32_t first_row, uint32_t rows, u32x4 (&v)[2]) {
  typedef const volatile __attribute__((address_space(1))) u32x4 global_u32x4_now;
  const char* at = static_cast<const char*>(data) + static_cast<size_t>(first_row) * width;
  if constexpr (AHEAD) v[0] = *(global_u32x4_now*)at; else v[0] = *(const global_u32x4*)at;
  v[1] = u32x4{0, 0, 0, 0};
  if (width == 2 && first_row + 8 < rows) {
    if constexpr (AHEAD) v[1] = *(global_u32x4_now*)(at + 16); else v[1] = *(const global_u32x4*)(at + 16);
  }
}

constexpr uint32_t SD_THREADS = 512;                        // one workgroup per chunk, two workgroups per CU (each takes half of the CU's LDS)
constexpr uint32_t SD_STEPS = 8;                            // steps of 16 rows per lane and span
constexpr uint32_t SD_SPAN = SD_THREADS * SD_ROWS * SD_STEPS - SD_ROWS;   // 65520: where the next span of a chunk with more than 65535 rows starts (a span has fewer than 2^16 rows: see phase 2)
constexpr uint32_t SD_DICT_BYTES = 64 * 1024;               // LDS for a window of a wide column's (dense group, value id) counters, 16 bits each
constexpr uint32_t SD_WINDOW_IDS = SD_DICT_BYTES / 2 / SD_DENSE;   // 8192 value ids per window
__host__ __device__ constexpr size_t sd_lds_bytes() { return SD_DICT_BYTES + size_t{SD_NARROW} * SD_DENSE * 256 * 4 + size_t{SD_NARROW} * 256 * 8; }

// The 2-byte value ids' dictionary does not fit a CU's L1 (240 KB for l_extendedprice), and 60 M gathers that each pull a 128-byte line
// out of the L2 for four bytes were 700 of this kernel's first version's 980 us (and what bounds aggregate_rows).  Staging the
// dictionary in LDS windows and gathering there cut that to 270 us (four windows x a compare, an LDS read and four select-adds in double
// precision per row).  What runs now does no gather at all: see phase 2.  The rows' dense groups are kept in registers (four bits per
// row) between the windows; only the 2-byte ids are read again, out of the L2.
__global__ __launch_bounds__(SD_THREADS, 4) void aggregate_small_domain(AggArgs a, SmallDomainPlan plan, uint32_t n_chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sd_smem[];
  uint32_t (*s_hist)[SD_DENSE][256] = reinterpret_cast<uint32_t (*)[SD_DENSE][256]>(sd_smem + SD_DICT_BYTES);                   // [narrow column][dense group][value id] rows
  double (*s_dict)[256] = reinterpret_cast<double (*)[256]>(sd_smem + SD_DICT_BYTES + size_t{SD_NARROW} * SD_DENSE * 256 * 4);   // narrow columns' dictionaries as doubles
  __shared__ uint32_t s_dense_of_code[SD_CODES];               // 0xFF unassigned, 0xFE being assigned, else the dense index (may be >= SD_DENSE: a shared-cell group)
  __shared__ uint32_t s_code_of_dense[SD_CODES];
  __shared__ uint32_t s_n_dense;
  __shared__ __attribute__((aligned(8))) uint32_t s_dense_map[4];   // [0..1] sixteen nibbles: dense index of code c | [2] bit c: the nibble is valid
  __shared__ double s_sum[SD_CODES][SD_COLUMNS];               // per dense index (all of them) and column
  __shared__ uint32_t s_nonnull[SD_CODES][SD_COLUMNS];
  __shared__ uint32_t s_rows[SD_CODES], s_first[SD_CODES], s_last[SD_CODES];
  __shared__ uint32_t s_spare[64];                             // where rows without a dense group count (never read)
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const uint64_t chunk_base = a.row_base[chunk];

  // ---- descriptors ---------------------------------------------------------------------------------------------------------
  const void* key_data[MAX_GROUPBY];
  uint32_t key_stride[MAX_GROUPBY], key_size[MAX_GROUPBY];
  uint32_t rows_in_chunk = 0, product = 1;
#pragma unroll
  for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
    key_data[g] = nullptr;
    key_stride[g] = 0;
    key_size[g] = 0;
    if (g < a.n_groupby) {
      const DevSegment seg = a.groupby[g].segments[chunk];
      key_data[g] = seg.data;
      key_size[g] = seg.aux_size;
      key_stride[g] = product;
      product *= seg.aux_size + 1;   // + 1: the NULL value id
      rows_in_chunk = seg.size;
    }
  }
  const void* column_data[SD_COLUMNS];
  const void* column_dictionary[SD_COLUMNS];
  uint32_t column_size[SD_COLUMNS], column_type[SD_COLUMNS];
#pragma unroll
  for (uint32_t c = 0; c < SD_COLUMNS; ++c) {
    column_data[c] = column_dictionary[c] = nullptr;
    column_size[c] = column_type[c] = 0;
    if (c < plan.n_columns) {
      const DevSegment seg = plan.column[c][chunk];
      column_data[c] = seg.data;
      column_dictionary[c] = seg.aux;
      column_size[c] = seg.aux_size;
      column_type[c] = seg.data_type;
      if (a.n_groupby == 0) rows_in_chunk = seg.size;
    }
  }
  for (uint32_t i = tid; i < SD_NARROW * SD_DENSE * 256; i += SD_THREADS) (&s_hist[0][0][0])[i] = 0;
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) {
    if (c >= plan.n_narrow || tid >= 256) continue;   // (a narrow column's dictionary has at most 255 entries)
    double value = 0.0;
    if (tid < column_size[c]) value = column_type[c] == HY_TYPE_FLOAT ? static_cast<double>(((const global_f32*)column_dictionary[c])[tid]) : ((const global_f64*)column_dictionary[c])[tid];
    s_dict[c][tid] = value;
  }
  if (tid < SD_CODES) {
    s_dense_of_code[tid] = 0xFFu;
    s_code_of_dense[tid] = 0;
    s_rows[tid] = 0;
    s_first[tid] = 0xFFFFFFFFu;
    s_last[tid] = 0;
    for (uint32_t c = 0; c < SD_COLUMNS; ++c) { s_sum[tid][c] = 0.0; s_nonnull[tid][c] = 0; }
  }
  if (tid == 0) s_n_dense = 0;
  if (tid < 4) s_dense_map[tid] = 0;
  __syncthreads();

  const uint32_t n_wide = plan.n_columns - plan.n_narrow;
  uint32_t rows_of[SD_DENSE], first_of[SD_DENSE], last_of[SD_DENSE];
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) { rows_of[k] = 0; first_of[k] = 0xFFFFFFFFu; last_of[k] = 0; }
  uint32_t span_end = 0;
#pragma unroll 1
  for (uint32_t span = 0; span < rows_in_chunk; span = span_end) {
    span_end = rows_in_chunk - span <= 65535u ? rows_in_chunk : span + SD_SPAN;   // a Hyrise chunk (at most 65535 rows) is one span
    // ---- phase 1: the rows' groups (kept for phase 2), their bookkeeping, the 1-byte columns' histograms ---------------------------------
    uint64_t dense0 = 0, dense1 = 0, dense2 = 0, dense3 = 0, dense4 = 0, dense5 = 0, dense6 = 0, dense7 = 0;   // (named registers, selected by the step: an array indexed by a loop counter would live in scratch memory)
    static_assert(SD_STEPS == 8, "dense0 .. dense7");
#pragma unroll 1
    for (uint32_t step = 0; step < SD_STEPS; ++step) {
      const uint32_t first = span + (step * SD_THREADS + tid) * SD_ROWS;
      const uint32_t load_row = first < span_end ? first : span;   // (a lane without rows reads the chunk's first ids)
      u32x4 key_ids[MAX_GROUPBY][2], narrow_ids[SD_NARROW];
#pragma unroll
      for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
        key_ids[g][0] = key_ids[g][1] = u32x4{0, 0, 0, 0};
        if (g < a.n_groupby) sd_load_ids(key_data[g], plan.key_width[g], load_row, rows_in_chunk, key_ids[g]);
      }
#pragma unroll
      for (uint32_t c = 0; c < SD_NARROW; ++c) {
        narrow_ids[c] = u32x4{0, 0, 0, 0};
        if (c < plan.n_narrow) narrow_ids[c] = *(const global_u32x4*)(static_cast<const char*>(column_data[c]) + load_row);
      }
      uint64_t codes = 0;   // the rows' codes, four bits each
#pragma unroll
      for (uint32_t j = 0; j < SD_ROWS; ++j) {
        uint32_t code = 0;
#pragma unroll
        for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
          if (g < a.n_groupby) {
            const uint32_t id = sd_id(key_ids[g], plan.key_width[g], j);
            code += (id < key_size[g] ? id : key_size[g]) * key_stride[g];
          }
        }
        codes |= static_cast<uint64_t>(code & 0xFu) << (4 * j);
      }
      // Dense indices of the codes.  The map code -> dense index is sixteen nibbles: one 64-bit word (and sixteen "assigned" bits) read once
      // per step, a shift and a mask per row.  Only while the chunk still meets new codes -- its first rows -- a lane claims the code's
      // entry (lanes of a wave run in lockstep: the claimant never waits, the others look again) and the word is read again.
      uint64_t dense = 0;
      if (plan.debug & 8) dense = codes & 0x3333333333333333ull;
      else {
        uint64_t map;
        uint32_t assigned = *reinterpret_cast<volatile uint32_t*>(&s_dense_map[2]);
        uint32_t wanted = 0;   // codes of this lane's rows
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          if (first + j < span_end) wanted |= 1u << (static_cast<uint32_t>(codes >> (4 * j)) & 0xFu);
        }
        while (__any((wanted & ~assigned) != 0)) {
          uint32_t missing = wanted & ~assigned;
          while (missing) {
            const uint32_t code = __ffs(missing) - 1;
            missing &= missing - 1;
            const uint32_t seen = atomicCAS(&s_dense_of_code[code], 0xFFu, 0xFEu);
            if (seen == 0xFFu) {   // this lane enters the code
              const uint32_t d = atomicAdd(&s_n_dense, 1u);
              s_code_of_dense[d] = code;
              __hip_atomic_store(&s_dense_of_code[code], d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              atomicOr(&s_dense_map[code >> 3], d << (4 * (code & 7u)));
              __hip_atomic_fetch_or(&s_dense_map[2], 1u << code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          assigned = *reinterpret_cast<volatile uint32_t*>(&s_dense_map[2]);   // (a code another lane or wave is still entering shows up a few instructions later: look again)
        }
        map = *reinterpret_cast<volatile uint64_t*>(&s_dense_map[0]);   // (entries are written before their assigned bit)
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t code = static_cast<uint32_t>(codes >> (4 * j)) & 0xFu;
          const uint32_t d = static_cast<uint32_t>(map >> (4 * code)) & 0xFu;
          dense |= static_cast<uint64_t>(first + j < span_end ? d : 0xFu) << (4 * j);   // (rows that do not exist match no group)
        }
      }
      if (step == 0) dense0 = dense; else if (step == 1) dense1 = dense; else if (step == 2) dense2 = dense; else if (step == 3) dense3 = dense;
      else if (step == 4) dense4 = dense; else if (step == 5) dense5 = dense; else if (step == 6) dense6 = dense; else dense7 = dense;
      // rows, first and last row per dense group: nibble arithmetic on the sixteen dense indices (a zero nibble of dense ^ k * 0x1111...
      // is a row of group k; the classic zero-in-word test marks it in the nibble's top bit)
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        const uint64_t x = dense ^ (0x1111111111111111ull * k);
        const uint64_t low3 = (x & 0x7777777777777777ull) + 0x7777777777777777ull;      // top bit of a nibble: its low three bits are not all zero
        const uint64_t hits = ~(low3 | x) & 0x8888888888888888ull;                      // ... nor its top bit: the nibble is zero
        if (hits) {
          rows_of[k] += __popcll(hits);
          const uint32_t first_hit = (__ffsll(static_cast<long long>(hits)) - 1) >> 2, last_hit = (63 - __clzll(static_cast<long long>(hits))) >> 2;
          first_of[k] = min(first_of[k], first + first_hit);
          last_of[k] = first + last_hit;
        }
      }
      // a fifth, sixth ... group of this chunk: shared LDS cells, row by row (rare)
      if (__any((dense & 0xCCCCCCCCCCCCCCCCull) != 0)) {
#pragma unroll 1
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
          if (d < SD_DENSE || first + j >= span_end) continue;
          const uint32_t row = first + j;
          atomicAdd(&s_rows[d], 1u);
          atomicMin(&s_first[d], row);
          atomicMax(&s_last[d], row);
#pragma unroll
          for (uint32_t c = 0; c < SD_COLUMNS; ++c) {
            if (c >= plan.n_columns) continue;
            const uint32_t width = c < plan.n_narrow ? 1u : 2u;
            const char* ids = static_cast<const char*>(column_data[c]) + static_cast<size_t>(row) * width;
            const uint32_t id = width == 1 ? *reinterpret_cast<const uint8_t*>(ids) : *reinterpret_cast<const uint16_t*>(ids);
            if (id >= column_size[c]) continue;
            const double value = column_type[c] == HY_TYPE_FLOAT ? static_cast<double>(static_cast<const float*>(column_dictionary[c])[id]) : static_cast<const double*>(column_dictionary[c])[id];
            atomicAdd(&s_sum[d][c], value);
            atomicAdd(&s_nonnull[d][c], 1u);
          }
        }
      }
      // 1-byte value ids: count the row in the histogram of its (dense group, value id) -- NULL ids are counted like the others and
      // left out when the counts are weighted; rows of other groups and rows that do not exist count in a spare row
#pragma unroll
      for (uint32_t c = 0; c < SD_NARROW; ++c) {
        if (c >= plan.n_narrow || (plan.debug & 1)) continue;
        const u32x4 ids[2] = {narrow_ids[c], u32x4{0, 0, 0, 0}};
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
          const uint32_t id = sd_id(ids, 1u, j);
          atomicAdd(d < SD_DENSE ? &s_hist[c][d][id] : &s_spare[id & 63u], 1u);
        }
      }
    }
    // ---- phase 2: the 2-byte columns, one window of value ids at a time ------------------------------------------------------------------
    // The rows of a window are COUNTED per (dense group, value id) in 16-bit LDS counters -- one LDS atomic per row on a packed pair;
    // a span has fewer than 2^16 rows, so a counter cannot carry into its neighbour -- and the counts are weighted with the window's
    // dictionary entries, read once, coalesced: no gather anywhere, and no floating-point work per row.
#pragma unroll 1
    for (uint32_t w = 0; w < ((plan.debug & 2) ? 0u : n_wide); ++w) {
      const uint32_t c = plan.n_narrow + w;
      const void* data = w == 0 ? column_data[plan.n_narrow < SD_COLUMNS ? plan.n_narrow : 0] : column_data[plan.n_narrow + 1 < SD_COLUMNS ? plan.n_narrow + 1 : 0];
      const void* dictionary = w == 0 ? column_dictionary[plan.n_narrow < SD_COLUMNS ? plan.n_narrow : 0] : column_dictionary[plan.n_narrow + 1 < SD_COLUMNS ? plan.n_narrow + 1 : 0];
      const uint32_t size = w == 0 ? column_size[plan.n_narrow < SD_COLUMNS ? plan.n_narrow : 0] : column_size[plan.n_narrow + 1 < SD_COLUMNS ? plan.n_narrow + 1 : 0];
      const bool is_float = (w == 0 ? column_type[plan.n_narrow < SD_COLUMNS ? plan.n_narrow : 0] : column_type[plan.n_narrow + 1 < SD_COLUMNS ? plan.n_narrow + 1 : 0]) == HY_TYPE_FLOAT;
      uint32_t* s_window = reinterpret_cast<uint32_t*>(sd_smem);   // [SD_DENSE][SD_WINDOW_IDS / 2] pairs of 16-bit counters
      double acc[SD_DENSE] = {0.0, 0.0, 0.0, 0.0};
      uint32_t counted[SD_DENSE] = {0, 0, 0, 0};   // non-NULL inputs per dense group
#pragma unroll 1
      for (uint32_t origin = 0; origin < size; origin += SD_WINDOW_IDS) {
        const uint32_t here = size - origin < SD_WINDOW_IDS ? size - origin : SD_WINDOW_IDS;
        __syncthreads();   // (the previous window has been weighted)
        for (uint32_t i = tid; i < SD_DICT_BYTES / 16; i += SD_THREADS) reinterpret_cast<u32x4*>(sd_smem)[i] = u32x4{0, 0, 0, 0};
        __syncthreads();
        // four steps' ids in flight at once (one after the other, sixty-four dependent L2 round trips per chunk -- eight windows x eight
        // steps -- bound this phase)
#pragma unroll 1
        for (uint32_t group = 0; group < SD_STEPS; group += 4) {
          u32x4 ids[4][2];
#pragma unroll
          for (uint32_t s4 = 0; s4 < 4; ++s4) {
            const uint32_t first = span + ((group + s4) * SD_THREADS + tid) * SD_ROWS;
            sd_load_ids(data, 2u, first < span_end ? first : span, rows_in_chunk, ids[s4]);
          }
#pragma unroll
          for (uint32_t s4 = 0; s4 < 4; ++s4) {
            const uint64_t dense = group == 0 ? (s4 == 0 ? dense0 : s4 == 1 ? dense1 : s4 == 2 ? dense2 : dense3) : (s4 == 0 ? dense4 : s4 == 1 ? dense5 : s4 == 2 ? dense6 : dense7);
            if (plan.debug & 4) { acc[0] += static_cast<double>(ids[s4][0].x ^ ids[s4][1].w); continue; }
#pragma unroll
            for (uint32_t j = 0; j < SD_ROWS; ++j) {
              const uint32_t local = sd_id(ids[s4], 2u, j) - origin;
              const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
              // (unsigned: ids below the window wrap around; the NULL id, size, is in no window; rows of a fifth ... group and rows that
              // do not exist have d >= 4)
              if (local < here && d < SD_DENSE) atomicAdd(&s_window[d * (SD_WINDOW_IDS / 2) + (local >> 1)], 1u << (16 * (local & 1)));
            }
          }
        }
        __syncthreads();
        // weights: thread t takes the window's entries t, t + SD_THREADS, ... -- all of its dictionary loads in flight at once (they come
        // from HBM: one after the other, sixteen round trips per window were most of this kernel)
#pragma unroll 1
        for (uint32_t batch = 0; batch < SD_WINDOW_IDS / SD_THREADS; batch += 8) {
          if (batch * SD_THREADS >= here) break;
          double value[8];
#pragma unroll
          for (uint32_t n = 0; n < 8; ++n) {
            const uint32_t i = (batch + n) * SD_THREADS + tid;
            const uint32_t at = origin + (i < here ? i : 0u);
            value[n] = is_float ? static_cast<double>(((const global_f32*)dictionary)[at]) : ((const global_f64*)dictionary)[at];
          }
#pragma unroll
          for (uint32_t n = 0; n < 8; ++n) {
            const uint32_t i = (batch + n) * SD_THREADS + tid;
#pragma unroll
            for (uint32_t k = 0; k < SD_DENSE; ++k) {
              const uint32_t count = i < here ? (s_window[k * (SD_WINDOW_IDS / 2) + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu : 0u;
              acc[k] += static_cast<double>(count) * value[n];
              counted[k] += count;
            }
          }
        }
      }
      __syncthreads();
      // the column's sums -> LDS cells (a wave reduction each, then one LDS atomic per wave)
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        const uint64_t sum = wave_reduce_to_lane63(static_cast<uint64_t>(__double_as_longlong(acc[k])), 0ull, [](uint64_t x, uint64_t y) {
          return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) + __longlong_as_double(static_cast<long long>(y))));
        });
        const uint32_t inputs = wave_reduce_u32_to_lane63(counted[k], 0u, false, false);
        if (lane == 63) {
          atomicAdd(&s_sum[k][c], __longlong_as_double(static_cast<long long>(sum)));
          atomicAdd(&s_nonnull[k][c], inputs);
        }
      }
    }
  }
  __syncthreads();

  // ---- the chunk's groups --------------------------------------------------------------------------------------------------
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) {
    const uint32_t rows = wave_reduce_u32_to_lane63(rows_of[k], 0u, false, false);
    const uint32_t first = wave_reduce_u32_to_lane63(first_of[k], 0xFFFFFFFFu, true, false), last = wave_reduce_u32_to_lane63(last_of[k], 0u, false, true);
    if (lane == 63 && rows) {
      atomicAdd(&s_rows[k], rows);
      atomicMin(&s_first[k], first);
      atomicMax(&s_last[k], last);
    }
  }
  // histograms x dictionaries: thread = value id
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) {
    if (c >= plan.n_narrow || tid >= 256) continue;
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) {
      const uint32_t count = tid < column_size[c] ? s_hist[c][k][tid] : 0u;
      const uint64_t sum = wave_reduce_to_lane63(static_cast<uint64_t>(__double_as_longlong(static_cast<double>(count) * s_dict[c][tid])), 0ull, [](uint64_t x, uint64_t y) {
        return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) + __longlong_as_double(static_cast<long long>(y))));
      });
      const uint32_t total = wave_reduce_u32_to_lane63(count, 0u, false, false);
      if (lane == 63 && total) {
        atomicAdd(&s_sum[k][c], __longlong_as_double(static_cast<long long>(sum)));
        atomicAdd(&s_nonnull[k][c], total);
      }
    }
  }
  __syncthreads();
  // merge into the global table: thread = dense index
  const uint32_t n_dense = s_n_dense;
  if (tid >= n_dense || s_rows[tid] == 0) return;
  if (__hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const uint32_t code = s_code_of_dense[tid];
  const uint32_t words = a.n_groupby + 1;
  uint64_t tuple[MAX_GROUPBY + 1];
  tuple[0] = 0;
#pragma unroll
  for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
    tuple[g + 1] = 0;
    if (g >= a.n_groupby) continue;
    const uint32_t id = (code / key_stride[g]) % (key_size[g] + 1);
    if (id >= key_size[g]) { tuple[0] |= 1ull << g; continue; }
    const DevSegment seg = a.groupby[g].segments[chunk];
    uint64_t bits;
    switch (seg.data_type) {
      case HY_TYPE_INT: bits = static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int32_t*>(seg.aux)[id])); break;
      case HY_TYPE_LONG: bits = static_cast<const uint64_t*>(seg.aux)[id]; break;
      case HY_TYPE_FLOAT: bits = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<const float*>(seg.aux)[id]))); break;
      default: bits = static_cast<const uint64_t*>(seg.aux)[id]; break;
    }
    if (a.groupby[g].is_float && __longlong_as_double(static_cast<long long>(bits)) == 0.0) bits = 0;
    tuple[g + 1] = bits;
  }
  const uint32_t gslot = global_slot(a, tuple, words, hash_tuple_in_registers(tuple, words));
  if (gslot == 0xFFFFFFFFu) { *a.overflow = 1; return; }
  atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(chunk_base + s_first[tid]));
  atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(chunk_base + s_last[tid]));
  for (uint32_t g = 0; g < a.n_aggregates; ++g) {
    const uint32_t c = plan.column_of_aggregate[g];
    if (c == 0xFFFFFFFFu) { merge_global(a, gslot, g, 0, s_rows[tid]); continue; }
    merge_global(a, gslot, g, static_cast<uint64_t>(__double_as_longlong(s_sum[tid][c])), s_nonnull[tid][c]);
  }
}
