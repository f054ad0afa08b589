// aggregate_small.hpp -- AggregateHash for the TPC-H Q1 shape (included by aggregate.hip, inside namespace hy).
//
// What it replaces (reference, CPU): the row loop of AggregateHash::_aggregate for a handful of groups
// (operators/aggregate_hash.cpp:317-403 get_or_add_result, :605-655 _aggregate_segment, :1016-1176) -- config 4 of BASELINE.json:
// GROUP BY l_returnflag, l_linestatus (dictionary segments, a few distinct values) with SUM / AVG / COUNT over DictionarySegment<float>
// columns.  aggregate_rows handles every encoding x type x function in one 150 KB body at 167 registers and ran this shape at 11 %
// of the HBM roofline; this kernel takes the shape and nothing else:
//   * at most two GROUP BY columns, each a dictionary segment with 1-byte value ids in every chunk, the product of their (dictionary
//     size + 1) at most 16: a row's group is the mixed-radix CODE of its value ids; the first four codes a chunk meets are its DENSE
//     groups (Q1 has four),
//   * every aggregate is SUM / AVG / COUNT over a dictionary-encoded float / double column with 1- or 2-byte value ids (or COUNT(*)),
//   * 1-byte value ids (l_quantity, l_discount): the rows are COUNTED per (value id, dense group) in an LDS histogram -- one LDS
//     atomic per row and column, no dictionary gather at all -- and the counts are weighted with the dictionary once per chunk
//     (the double sums are exact for these columns' products count x value in any order),
//   * 2-byte value ids (l_extendedprice, 240 KB of dictionary per chunk): counted as well, in one 4-bit counter per (value id, dense
//     group) -- 128 KB of LDS -- and weighted with the dictionary, which is read once, coalesced (see the kernel's comment).
// One workgroup of 1024 threads per chunk, sixteen consecutive rows per lane and step (16-byte loads of 1-byte ids, two for 2-byte ids),
// a step's loads issued one step ahead.  Rows of a fifth, sixth ... group of a chunk take LDS atomics on shared cells.  The chunk's
// groups are merged into the global table like aggregate_rows' (global_slot / merge_global): result order, representative rows and
// values are those of the generic kernel.
// SUM / AVG: double additions in a different order than the reference's row loop -- the stated 1e-9 relative tolerance.
#pragma once

constexpr uint32_t SD_CODES = 16;       // product of (dictionary size + 1) over the GROUP BY columns, at most
constexpr uint32_t SD_DENSE = 4;        // groups of a chunk with register / histogram accumulators
constexpr uint32_t SD_COLUMNS = 4;      // distinct aggregate input columns
constexpr uint32_t SD_NARROW = 2;       // ... of which with 1-byte value ids, at most
constexpr uint32_t SD_WIDE = 1;         // ... and with 2-byte value ids
constexpr uint32_t SD_JOINT_CELLS = 1024; // (value id, value id) pairs of two 1-byte columns counted in one histogram (the two single histograms' LDS)
constexpr uint32_t SD_ROWS = 16;        // consecutive rows of a lane per step
constexpr uint32_t SD_KEYS = 2;         // GROUP BY columns, at most (1-byte value ids: their dictionaries have fewer than sixteen entries)
typedef __attribute__((address_space(1))) float global_f32;    // (pointers read from a segment descriptor are generic to the compiler: flat loads, which also count on lgkmcnt)
typedef __attribute__((address_space(1))) double global_f64;

struct SmallDomainPlan {
  uint32_t n_columns, n_narrow;                   // distinct input columns; the first n_narrow have 1-byte value ids, the others 2-byte ones
  const DevSegment* column[SD_COLUMNS];
  uint32_t column_of_aggregate[MAX_AGGREGATES];   // 0xFFFFFFFF: COUNT(*)
  uint32_t joint;                                 // two 1-byte columns whose (dictionary size + 1)s multiply to at most SD_JOINT_CELLS in every chunk: ONE histogram over the pair
  uint32_t debug;                                 // HY_AGG_SMALL_DEBUG (timing experiments, wrong results): 1 no histograms, 2 no 2-byte columns, 8 no dense lookup
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

constexpr uint32_t SD_THREADS = 1024;                       // one workgroup per chunk and CU (it takes most of the CU's LDS): 16 waves
constexpr uint32_t SD_STEPS = 4;                            // steps of 16 rows per lane and span
constexpr uint32_t SD_SPAN = SD_THREADS * SD_ROWS * SD_STEPS;   // 65536 rows: a Hyrise chunk (at most 65535 rows) is one span
constexpr uint32_t SD_WIDE_BYTES = 128 * 1024;              // one 16-bit cell per value id of a wide column: four 4-bit counters, one per dense group
constexpr uint32_t SD_COPIES = 2;                           // copies of the narrow columns' histograms (even / odd lanes)
__host__ __device__ constexpr size_t sd_lds_bytes() { return SD_WIDE_BYTES + size_t{SD_COPIES} * SD_NARROW * 256 * SD_DENSE * 4 + size_t{SD_NARROW} * 256 * 8; }

// What a lane asks for per step: sixteen consecutive rows of every column (16 bytes of 1-byte ids, 2 x 16 bytes of 2-byte ids).
struct SdStep {
  u32x4 key[SD_KEYS];
  u32x4 narrow[SD_NARROW];
  u32x4 wide[2];
};

// History of the 2-byte column (l_extendedprice, 240 KB of dictionary per chunk -- more than a CU's L1, and 60 M gathers that each pull
// a 128-byte line out of the L2 for four bytes were 700 of the first version's 980 us and are what bounds aggregate_rows): gathers from
// dictionary windows staged in LDS, 550 us; rows COUNTED per (dense group, value id) in 16-bit LDS counters, eight windows of 8192 ids,
// 520 us -- 316 of them the windows: 64 dependent round trips per chunk (eight windows x (zero, ids from the L2 twice, dictionary from
// HBM twice)) with two workgroups per CU to hide them.  Now ONE window: a 4-bit counter per (dense group, value id) -- 128 KB of LDS
// for 65536 ids -- so a row costs one LDS atomic and the chunk is read once; the dictionary is read once, coalesced, when the counts
// are weighted.  A counter that meets a sixteenth row carries into its neighbour; every carry lowers the sum of all counters, which is
// compared with the rows counted in registers: a chunk that fails the comparison (sixteen rows of one group with one price in 65535
// rows) raises FLAG_SMALL_REFUSED and the host runs aggregate_rows instead.
__global__ __launch_bounds__(SD_THREADS) void aggregate_small_domain(AggArgs a, SmallDomainPlan plan, uint32_t n_chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sd_smem[];
  uint32_t* s_wide = reinterpret_cast<uint32_t*>(sd_smem);                                                        // [32768] two 16-bit cells each
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(sd_smem + SD_WIDE_BYTES);                                         // [copy][narrow column][value id][dense group] rows
  double (*s_dict)[256] = reinterpret_cast<double (*)[256]>(sd_smem + SD_WIDE_BYTES + size_t{SD_COPIES} * SD_NARROW * 256 * SD_DENSE * 4);   // narrow columns' dictionaries as doubles
  __shared__ uint32_t s_dense_of_code[SD_CODES];               // 0xFF unassigned, 0xFE being assigned, else the dense index (may be >= SD_DENSE: a shared-cell group)
  __shared__ uint32_t s_code_of_dense[SD_CODES];
  __shared__ uint32_t s_n_dense;
  __shared__ uint32_t s_check;                                 // counters summed - rows counted (wide columns): not zero = a counter overflowed
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
  const void* key_data[SD_KEYS];
  uint32_t key_stride[SD_KEYS], key_size[SD_KEYS];
  uint32_t rows_in_chunk = 0, product = 1;
#pragma unroll
  for (uint32_t g = 0; g < SD_KEYS; ++g) {
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
  const uint32_t n_wide = plan.n_columns - plan.n_narrow;
  const void* wide_data = nullptr;       // the 2-byte column (selected with static indices: an array indexed at run time would live in scratch memory)
  const void* wide_dictionary = nullptr;
  uint32_t wide_size = 0;
  bool wide_is_float = false;
#pragma unroll
  for (uint32_t c = 0; c < SD_COLUMNS; ++c) {
    if (n_wide && c == plan.n_narrow) { wide_data = column_data[c]; wide_dictionary = column_dictionary[c]; wide_size = column_size[c]; wide_is_float = column_type[c] == HY_TYPE_FLOAT; }
  }

  // The loads of a step: issued one step ahead of their use (volatile: the compiler would sink them down to it).
  auto load_step = [&](uint32_t span, uint32_t span_end, uint32_t step, SdStep& s) {
    typedef const volatile __attribute__((address_space(1))) u32x4 global_u32x4_now;
    const uint32_t first = span + (step * SD_THREADS + tid) * SD_ROWS;
    const uint32_t row = first < span_end ? first : span;   // (a lane without rows reads the span's first ids)
#pragma unroll
    for (uint32_t g = 0; g < SD_KEYS; ++g) {
      s.key[g] = u32x4{0, 0, 0, 0};
      if (g < a.n_groupby) s.key[g] = *(global_u32x4_now*)(static_cast<const char*>(key_data[g]) + row);
    }
#pragma unroll
    for (uint32_t c = 0; c < SD_NARROW; ++c) {
      s.narrow[c] = u32x4{0, 0, 0, 0};
      if (c < plan.n_narrow) s.narrow[c] = *(global_u32x4_now*)(static_cast<const char*>(column_data[c]) + row);
    }
    s.wide[0] = s.wide[1] = u32x4{0, 0, 0, 0};
    if (n_wide) {
      const char* at = static_cast<const char*>(wide_data) + size_t{row} * 2;
      s.wide[0] = *(global_u32x4_now*)at;
      if (row + 8 < rows_in_chunk) s.wide[1] = *(global_u32x4_now*)(at + 16);   // (a load starts at an existing row: it ends inside the padding every uploaded buffer has)
    }
  };

  for (uint32_t i = tid; i < SD_COPIES * SD_NARROW * 256 * SD_DENSE; i += SD_THREADS) s_hist[i] = 0;
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
  if (tid == 0) { s_n_dense = 0; s_check = 0; }
  if (tid < 4) s_dense_map[tid] = 0;

  uint32_t rows_of[SD_DENSE], first_of[SD_DENSE], last_of[SD_DENSE];
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) { rows_of[k] = 0; first_of[k] = 0xFFFFFFFFu; last_of[k] = 0; }
  const uint32_t copy = lane & (SD_COPIES - 1);
  uint32_t span_end = 0;
#pragma unroll 1
  for (uint32_t span = 0; span < rows_in_chunk; span = span_end) {
    span_end = rows_in_chunk - span <= SD_SPAN ? rows_in_chunk : span + SD_SPAN;
    SdStep current;
    load_step(span, span_end, 0, current);
    if (span != 0) __syncthreads();   // (the span before has been weighted: its counters may go)
    if (n_wide) {
      for (uint32_t i = tid; i < SD_WIDE_BYTES / 16; i += SD_THREADS) reinterpret_cast<u32x4*>(sd_smem)[i] = u32x4{0, 0, 0, 0};
    }
    __syncthreads();   // (the tables are set up; an earlier span's counters have been weighted)
    uint32_t wide_rows = 0;   // rows this lane counted into the wide column's counters
    // ---- the rows: groups, bookkeeping, histograms of the 1-byte columns, counters of the first 2-byte column ---------------------------
#pragma unroll 1
    for (uint32_t step = 0; step < SD_STEPS; ++step) {
      const uint32_t first = span + (step * SD_THREADS + tid) * SD_ROWS;
      SdStep next = current;
      if (step + 1 < SD_STEPS && span + (step + 1) * SD_THREADS * SD_ROWS < span_end) load_step(span, span_end, step + 1, next);
      uint64_t codes = 0;   // the rows' codes, four bits each
#pragma unroll
      for (uint32_t j = 0; j < SD_ROWS; ++j) {
        uint32_t code = 0;
#pragma unroll
        for (uint32_t g = 0; g < SD_KEYS; ++g) {   // (a GROUP BY column that is not there: ids 0, stride 0)
          const u32x4 ids[2] = {current.key[g], u32x4{0, 0, 0, 0}};
          const uint32_t id = sd_id(ids, 1u, j);
          code += (id < key_size[g] ? id : key_size[g]) * key_stride[g];
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
      // 1-byte value ids: count the row in the histogram of its (value id, dense group) -- NULL ids are counted like the others and
      // left out when the counts are weighted; rows of other groups and rows that do not exist count in a spare row.  Cells of one
      // value id are neighbours (a column with eleven values, l_discount, would otherwise meet in eleven of the LDS's banks) and even
      // and odd lanes have their own copy.
      if (plan.joint && !(plan.debug & 1)) {
        // Two 1-byte columns (l_quantity x l_discount: 51 x 12 pairs): the row counts ONCE, in the cell of its pair of value ids -- one LDS
        // atomic instead of two; the pair counts are summed into each column's histogram when the chunk is done.
        const u32x4 first_ids[2] = {current.narrow[0], u32x4{0, 0, 0, 0}}, second_ids[2] = {current.narrow[1], u32x4{0, 0, 0, 0}};
        const uint32_t second_domain = column_size[1] + 1;
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
          const uint32_t id0 = sd_id(first_ids, 1u, j), id1 = sd_id(second_ids, 1u, j);
          const uint32_t cell = (id0 < column_size[0] ? id0 : column_size[0]) * second_domain + (id1 < column_size[1] ? id1 : column_size[1]);
          atomicAdd(d < SD_DENSE ? &s_hist[cell * SD_DENSE + d] : &s_spare[cell & 63u], 1u);
        }
      } else {
#pragma unroll
      for (uint32_t c = 0; c < SD_NARROW; ++c) {
        if (c >= plan.n_narrow || (plan.debug & 1)) continue;
        const u32x4 ids[2] = {current.narrow[c], u32x4{0, 0, 0, 0}};
        uint32_t* cells = s_hist + (copy * SD_NARROW + c) * 256 * SD_DENSE;
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
          const uint32_t id = sd_id(ids, 1u, j);
          atomicAdd(d < SD_DENSE ? &cells[id * SD_DENSE + d] : &s_spare[id & 63u], 1u);
        }
      }
      }
      // the first 2-byte column: one 4-bit counter per (value id, dense group); NULL ids (the dictionary's size) are not counted
      if (n_wide && !(plan.debug & 2)) {
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
          const uint32_t id = sd_id(current.wide, 2u, j);
          if (d < SD_DENSE && id < wide_size) {
            atomicAdd(&s_wide[id >> 1], 1u << (16 * (id & 1) + 4 * d));
            wide_rows += 1;
          }
        }
      }
      current = next;
    }
    // ---- the 2-byte columns' counters x their dictionaries -------------------------------------------------------------------------------
#pragma unroll 1
    for (uint32_t w = 0; w < ((plan.debug & 2) ? 0u : n_wide); ++w) {
      const uint32_t c = plan.n_narrow + w;
      const void* dictionary = wide_dictionary;
      const uint32_t size = wide_size;
      const bool is_float = wide_is_float;
      double acc[SD_DENSE] = {0.0, 0.0, 0.0, 0.0};
      uint32_t counted[SD_DENSE] = {0, 0, 0, 0};   // non-NULL inputs per dense group
      // Thread t takes the value ids t, t + SD_THREADS, ...: 32 dictionary entries (16 for doubles) requested at once -- they come from
      // HBM, coalesced, and one request after the other, eight round trips per chunk, was a sixth of this kernel -- and the first batch
      // is requested BEFORE the barrier that ends the counting (the dictionary does not depend on it).
      bool counting_done = false;
      auto weigh = [&](uint32_t i, double value) {
        const uint32_t cell = i < size ? (s_wide[i >> 1] >> (16 * (i & 1))) & 0xFFFFu : 0u;
#pragma unroll
        for (uint32_t k = 0; k < SD_DENSE; ++k) {
          const uint32_t count = (cell >> (4 * k)) & 0xFu;
          acc[k] += static_cast<double>(count) * value;
          counted[k] += count;
        }
      };
      if (is_float) {
        constexpr uint32_t BATCH = 32;
#pragma unroll 1
        for (uint32_t batch = 0; batch < 65536 / SD_THREADS; batch += BATCH) {
          if (batch * SD_THREADS >= size) break;
          float value[BATCH];
#pragma unroll
          for (uint32_t n = 0; n < BATCH; ++n) {
            const uint32_t i = (batch + n) * SD_THREADS + tid;
            value[n] = ((const global_f32*)dictionary)[i < size ? i : 0u];
          }
          if (!counting_done) { __syncthreads(); counting_done = true; }
#pragma unroll
          for (uint32_t n = 0; n < BATCH; ++n) weigh((batch + n) * SD_THREADS + tid, static_cast<double>(value[n]));
        }
      } else {
        constexpr uint32_t BATCH = 16;
#pragma unroll 1
        for (uint32_t batch = 0; batch < 65536 / SD_THREADS; batch += BATCH) {
          if (batch * SD_THREADS >= size) break;
          double value[BATCH];
#pragma unroll
          for (uint32_t n = 0; n < BATCH; ++n) {
            const uint32_t i = (batch + n) * SD_THREADS + tid;
            value[n] = ((const global_f64*)dictionary)[i < size ? i : 0u];
          }
          if (!counting_done) { __syncthreads(); counting_done = true; }
#pragma unroll
          for (uint32_t n = 0; n < BATCH; ++n) weigh((batch + n) * SD_THREADS + tid, value[n]);
        }
      }
      if (!counting_done) __syncthreads();   // (an empty dictionary: every row NULL)
      // the column's sums -> LDS cells (a wave reduction each, then one LDS atomic per wave); counters summed against rows counted
      uint32_t all_counted = 0;
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        const uint64_t sum = wave_reduce_to_lane63(static_cast<uint64_t>(__double_as_longlong(acc[k])), 0ull, [](uint64_t x, uint64_t y) {
          return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) + __longlong_as_double(static_cast<long long>(y))));
        });
        const uint32_t inputs = wave_reduce_u32_to_lane63(counted[k], 0u, false, false);
        all_counted += inputs;
        if (lane == 63) {
          atomicAdd(&s_sum[k][c], __longlong_as_double(static_cast<long long>(sum)));
          atomicAdd(&s_nonnull[k][c], inputs);
        }
      }
      const uint32_t expected = wave_reduce_u32_to_lane63(wide_rows, 0u, false, false);
      if (lane == 63 && all_counted != expected) atomicAdd(&s_check, all_counted - expected);
    }
  }
  __syncthreads();
  if (tid == 0 && s_check != 0) __hip_atomic_store(&a.overflow[FLAG_SMALL_REFUSED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

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
      uint32_t count = 0;
      if (plan.joint) {   // the column's histogram = the pair histogram summed over the other column's value ids (its NULL id included)
        const uint32_t second_domain = column_size[1] + 1;
        if (c == 0) { for (uint32_t other = 0; other < second_domain && tid < column_size[0]; ++other) count += s_hist[(tid * second_domain + other) * SD_DENSE + k]; }
        else { for (uint32_t other = 0; other <= column_size[0] && tid < column_size[1]; ++other) count += s_hist[(other * second_domain + tid) * SD_DENSE + k]; }
      } else {
#pragma unroll
        for (uint32_t copy_index = 0; copy_index < SD_COPIES; ++copy_index) count += s_hist[((copy_index * SD_NARROW + c) * 256 + tid) * SD_DENSE + k];
      }
      if (tid >= column_size[c]) count = 0;
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
    if (g >= a.n_groupby || g >= SD_KEYS) continue;
    const uint32_t id = (code / key_stride[g < SD_KEYS ? g : 0]) % (key_size[g < SD_KEYS ? g : 0] + 1);
    if (id >= key_size[g < SD_KEYS ? g : 0]) { tuple[0] |= 1ull << g; continue; }
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
