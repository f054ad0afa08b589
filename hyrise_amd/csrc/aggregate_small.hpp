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
//   * every aggregate is SUM / AVG / COUNT / MIN / MAX over a dictionary-encoded int / long / float / double column with 1- or 2-byte
//     value ids (or COUNT(*)): int and long columns sum in int64 (exact, AVG's double is that sum converted), MIN / MAX are the
//     smallest / largest value id a group COUNTED (dictionaries are sorted) looked up once per chunk,
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
typedef __attribute__((address_space(1))) int32_t global_i32;
typedef __attribute__((address_space(1))) int64_t global_i64;

__device__ __forceinline__ bool sd_is_integer(uint32_t data_type) { return data_type == HY_TYPE_INT || data_type == HY_TYPE_LONG; }
// entry i of a dictionary as the word the aggregates work on: the double's bits (float widened) or the int64's (int widened)
__device__ __forceinline__ uint64_t sd_dictionary_bits(const void* dictionary, uint32_t data_type, uint32_t i) {
  switch (data_type) {
    case HY_TYPE_INT: return static_cast<uint64_t>(static_cast<int64_t>(((const global_i32*)dictionary)[i]));
    case HY_TYPE_LONG: return static_cast<uint64_t>(((const global_i64*)dictionary)[i]);
    case HY_TYPE_FLOAT: return static_cast<uint64_t>(__double_as_longlong(static_cast<double>(((const global_f32*)dictionary)[i])));
    default: return static_cast<uint64_t>(__double_as_longlong(((const global_f64*)dictionary)[i]));
  }
}
// sum cell (LDS) += a column's contribution: int64 or double
__device__ __forceinline__ void sd_add(uint64_t* cell, uint64_t bits, bool integer) {
  if (integer) atomicAdd(reinterpret_cast<unsigned long long*>(cell), static_cast<unsigned long long>(bits));
  else atomicAdd(reinterpret_cast<double*>(cell), __longlong_as_double(static_cast<long long>(bits)));
}
__device__ __forceinline__ uint64_t sd_wave_sum(uint64_t bits, bool integer) {   // over the wave, result in lane 63
  if (integer) return wave_reduce_to_lane63(bits, 0ull, [](uint64_t x, uint64_t y) { return x + y; });
  return wave_reduce_to_lane63(bits, 0ull, [](uint64_t x, uint64_t y) {
    return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) + __longlong_as_double(static_cast<long long>(y))));
  });
}

struct SmallDomainPlan {
  uint32_t n_columns, n_narrow;                   // distinct input columns; the first n_narrow have 1-byte value ids, the others 2-byte ones
  const DevSegment* column[SD_COLUMNS];
  uint32_t column_of_aggregate[MAX_AGGREGATES];   // 0xFFFFFFFF: COUNT(*)
  uint32_t extremes;                              // bit c: a MIN / MAX reads column c (its groups' smallest / largest counted value ids are kept)
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

// The 2-byte column's counters x its dictionary (Stored: the dictionary's element type).  Thread t takes the value ids t, t + SD_THREADS,
// ...: 32 dictionary entries (16 of eight bytes) requested at once -- they come from HBM, coalesced, and one request after the other,
// eight round trips per chunk, was a sixth of the kernel -- and the first batch is requested BEFORE the barrier that ends the counting
// (the dictionary does not depend on it).  Out, per dense group: the sum (a double's bits; an int64 for int / long) and the inputs counted.
template <typename Stored>
__device__ __forceinline__ void sd_weigh_wide(const void* dictionary, uint32_t size, const uint32_t* s_wide, uint32_t tid, uint64_t (&sums)[SD_DENSE], uint32_t (&counted)[SD_DENSE]) {
  typedef const __attribute__((address_space(1))) Stored global_stored;
  constexpr bool INTEGER = std::is_integral<Stored>::value;
  typedef typename std::conditional<INTEGER, int64_t, double>::type Sum;
  constexpr uint32_t BATCH = sizeof(Stored) == 4 ? 32 : 16;
  Sum acc[SD_DENSE];
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) { acc[k] = 0; counted[k] = 0; }
  bool counting_done = false;
#pragma unroll 1
  for (uint32_t batch = 0; batch < 65536 / SD_THREADS; batch += BATCH) {
    if (batch * SD_THREADS >= size) break;
    Stored value[BATCH];
#pragma unroll
    for (uint32_t n = 0; n < BATCH; ++n) {
      const uint32_t i = (batch + n) * SD_THREADS + tid;
      value[n] = ((global_stored*)dictionary)[i < size ? i : 0u];
    }
    if (!counting_done) { __syncthreads(); counting_done = true; }
#pragma unroll
    for (uint32_t n = 0; n < BATCH; ++n) {
      const uint32_t i = (batch + n) * SD_THREADS + tid;
      const uint32_t cell = i < size ? (s_wide[i >> 1] >> (16 * (i & 1))) & 0xFFFFu : 0u;
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        const uint32_t count = (cell >> (4 * k)) & 0xFu;
        acc[k] += static_cast<Sum>(count) * static_cast<Sum>(value[n]);
        counted[k] += count;
      }
    }
  }
  if (!counting_done) __syncthreads();   // (an empty dictionary: every row NULL)
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) {
    if constexpr (INTEGER) sums[k] = static_cast<uint64_t>(acc[k]);
    else sums[k] = static_cast<uint64_t>(__double_as_longlong(acc[k]));
  }
}

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
  uint64_t (*s_dict)[256] = reinterpret_cast<uint64_t (*)[256]>(sd_smem + SD_WIDE_BYTES + size_t{SD_COPIES} * SD_NARROW * 256 * SD_DENSE * 4);   // narrow columns' dictionaries (sd_dictionary_bits)
  __shared__ uint32_t s_dense_of_code[SD_CODES];               // 0xFF unassigned, 0xFE being assigned, else the dense index (may be >= SD_DENSE: a shared-cell group)
  __shared__ uint32_t s_code_of_dense[SD_CODES];
  __shared__ uint32_t s_n_dense;
  __shared__ uint32_t s_check;                                 // counters summed - rows counted (wide columns): not zero = a counter overflowed
  __shared__ __attribute__((aligned(8))) uint32_t s_dense_map[4];   // [0..1] sixteen nibbles: dense index of code c | [2] bit c: the nibble is valid
  __shared__ uint64_t s_sum[SD_CODES][SD_COLUMNS];             // per dense index (all of them) and column: a double's bits, an int64 for int / long columns
  __shared__ uint32_t s_min_id[SD_CODES][SD_COLUMNS], s_max_id[SD_CODES][SD_COLUMNS];   // smallest / largest value id counted (plan.extremes)
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
  // What the row loop reads of the input columns stays in registers: the value ids and the dictionaries' sizes.  Dictionaries and types
  // are read from the descriptors again where the counts are weighted (the kernel runs at the scalar register limit as well, and scalar
  // registers that do not fit take vector registers).
  const void* narrow_data[SD_NARROW];
  uint32_t narrow_size[SD_NARROW];
  const uint32_t n_wide = plan.n_columns - plan.n_narrow;
  const void* wide_data = nullptr;
  uint32_t wide_size = 0;
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) { narrow_data[c] = nullptr; narrow_size[c] = 0; }
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW + SD_WIDE; ++c) {   // (static indices: an array indexed at run time would live in scratch memory)
    if (c >= plan.n_columns) continue;
    const DevSegment seg = plan.column[c][chunk];
    if (a.n_groupby == 0) rows_in_chunk = seg.size;
    if (c < SD_NARROW && c < plan.n_narrow) { narrow_data[c] = seg.data; narrow_size[c] = seg.aux_size; }
    else { wide_data = seg.data; wide_size = seg.aux_size; }
  }
  auto segment_again = [&](uint32_t c) {
    __asm__ volatile("" ::: "memory");   // (a load of its own: not the values of the loads above kept in registers)
    return plan.column[c][chunk];
  };

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
      if (c < plan.n_narrow) s.narrow[c] = *(global_u32x4_now*)(static_cast<const char*>(narrow_data[c]) + row);
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
    const DevSegment seg = plan.column[c][chunk];
    s_dict[c][tid] = tid < seg.aux_size ? sd_dictionary_bits(seg.aux, seg.data_type, tid) : 0ull;
  }
  if (tid < SD_CODES) {
    s_dense_of_code[tid] = 0xFFu;
    s_code_of_dense[tid] = 0;
    s_rows[tid] = 0;
    s_first[tid] = 0xFFFFFFFFu;
    s_last[tid] = 0;
    for (uint32_t c = 0; c < SD_COLUMNS; ++c) { s_sum[tid][c] = 0; s_nonnull[tid][c] = 0; s_min_id[tid][c] = 0xFFFFFFFFu; s_max_id[tid][c] = 0; }
  }
  if (tid == 0) { s_n_dense = 0; s_check = 0; }
  if (tid < 4) s_dense_map[tid] = 0;

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
    // Rows, first and last row of every dense group, per lane and SPAN, packed (the kernel runs at the register limit of a 1024-thread
    // workgroup and used to spill): a lane sees at most SD_ROWS x SD_STEPS = 64 rows of a span -- one byte per group --, and rows count
    // from the span's first -- sixteen bits each.  The spans' totals go to the LDS cells when the span is done.
    static_assert(SD_ROWS * SD_STEPS < 256 && SD_SPAN <= 65536 && SD_DENSE == 4, "rows of a lane per span: one byte per dense group; rows of a span: 16 bits");
    uint32_t rows_packed = 0;
    uint32_t extent_of[SD_DENSE];   // low half: first row (0xFFFF and no rows: none), high half: last row
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) extent_of[k] = 0xFFFFu;
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
          rows_packed += static_cast<uint32_t>(__popcll(hits)) << (8 * k);
          const uint32_t first_hit = (__ffsll(static_cast<long long>(hits)) - 1) >> 2, last_hit = (63 - __clzll(static_cast<long long>(hits))) >> 2;
          const uint32_t in_span = first - span;   // (a lane's rows ascend with the steps: the last hit so far is this step's)
          extent_of[k] = min(extent_of[k] & 0xFFFFu, in_span + first_hit) | (in_span + last_hit) << 16;
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
#pragma unroll 1
          for (uint32_t c = 0; c < plan.n_columns; ++c) {   // (a loop, not unrolled: this path is rare and must not cost the row loop registers)
            const uint32_t width = c < plan.n_narrow ? 1u : 2u;
            const DevSegment seg = segment_again(c);
            const char* ids = static_cast<const char*>(seg.data) + static_cast<size_t>(row) * width;
            const uint32_t id = width == 1 ? *reinterpret_cast<const uint8_t*>(ids) : *reinterpret_cast<const uint16_t*>(ids);
            if (id >= seg.aux_size) continue;
            sd_add(&s_sum[d][c], sd_dictionary_bits(seg.aux, seg.data_type, id), sd_is_integer(seg.data_type));
            atomicAdd(&s_nonnull[d][c], 1u);
            if ((plan.extremes >> c) & 1u) { atomicMin(&s_min_id[d][c], id); atomicMax(&s_max_id[d][c], id); }
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
        const uint32_t second_domain = narrow_size[1] + 1;
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t d = static_cast<uint32_t>(dense >> (4 * j)) & 0xFu;
          const uint32_t id0 = sd_id(first_ids, 1u, j), id1 = sd_id(second_ids, 1u, j);
          const uint32_t cell = (id0 < narrow_size[0] ? id0 : narrow_size[0]) * second_domain + (id1 < narrow_size[1] ? id1 : narrow_size[1]);
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
      const DevSegment wide = segment_again(c);
      const void* wide_dictionary = wide.aux;
      const uint32_t wide_type = wide.data_type;
      const bool integer = sd_is_integer(wide_type), track = (plan.extremes >> c) & 1u;
      uint64_t acc[SD_DENSE];                      // sums: doubles' bits, int64 for int / long columns
      uint32_t counted[SD_DENSE] = {0, 0, 0, 0};   // non-NULL inputs per dense group
      switch (wide_type) {
        case HY_TYPE_INT: sd_weigh_wide<int32_t>(wide_dictionary, wide_size, s_wide, tid, acc, counted); break;
        case HY_TYPE_LONG: sd_weigh_wide<int64_t>(wide_dictionary, wide_size, s_wide, tid, acc, counted); break;
        case HY_TYPE_FLOAT: sd_weigh_wide<float>(wide_dictionary, wide_size, s_wide, tid, acc, counted); break;
        default: sd_weigh_wide<double>(wide_dictionary, wide_size, s_wide, tid, acc, counted); break;
      }
      // the column's sums -> LDS cells (a wave reduction each, then one LDS atomic per wave); counters summed against rows counted
      uint32_t all_counted = 0;
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        const uint64_t sum = sd_wave_sum(acc[k], integer);
        const uint32_t inputs = wave_reduce_u32_to_lane63(counted[k], 0u, false, false);
        all_counted += inputs;
        if (lane == 63) {
          sd_add(&s_sum[k][c], sum, integer);
          atomicAdd(&s_nonnull[k][c], inputs);
        }
      }
      if (track) {   // MIN / MAX: the smallest and largest value id every dense group counted -- a pass of its own over the counters
        uint32_t low[SD_DENSE], high[SD_DENSE];   // (high: the largest value id + 1, 0 = none)
#pragma unroll
        for (uint32_t k = 0; k < SD_DENSE; ++k) { low[k] = 0xFFFFFFFFu; high[k] = 0; }
#pragma unroll 4
        for (uint32_t i = tid; i < wide_size; i += SD_THREADS) {   // (a thread's value ids ascend)
          const uint32_t cell = (s_wide[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
#pragma unroll
          for (uint32_t k = 0; k < SD_DENSE; ++k) {
            const bool counted_here = ((cell >> (4 * k)) & 0xFu) != 0;
            low[k] = counted_here ? min(low[k], i) : low[k];
            high[k] = counted_here ? i + 1 : high[k];
          }
        }
#pragma unroll
        for (uint32_t k = 0; k < SD_DENSE; ++k) {
          const uint32_t lowest = wave_reduce_u32_to_lane63(low[k], 0xFFFFFFFFu, true, false);
          const uint32_t highest = wave_reduce_u32_to_lane63(high[k], 0u, false, true);
          if (lane == 63 && highest) { atomicMin(&s_min_id[k][c], lowest); atomicMax(&s_max_id[k][c], highest - 1); }
        }
      }
      const uint32_t expected = wave_reduce_u32_to_lane63(wide_rows, 0u, false, false);
      if (lane == 63 && all_counted != expected) atomicAdd(&s_check, all_counted - expected);
    }
    // the span's dense groups: rows, first and last row -> the LDS cells
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) {
      const uint32_t mine = (rows_packed >> (8 * k)) & 0xFFu;
      const uint32_t rows = wave_reduce_u32_to_lane63(mine, 0u, false, false);
      const uint32_t first = wave_reduce_u32_to_lane63(mine ? span + (extent_of[k] & 0xFFFFu) : 0xFFFFFFFFu, 0xFFFFFFFFu, true, false);
      const uint32_t last = wave_reduce_u32_to_lane63(mine ? span + (extent_of[k] >> 16) : 0u, 0u, false, true);
      if (lane == 63 && rows) {
        atomicAdd(&s_rows[k], rows);
        atomicMin(&s_first[k], first);
        atomicMax(&s_last[k], last);
      }
    }
  }
  __syncthreads();
  if (tid == 0 && s_check != 0) __hip_atomic_store(&a.overflow[FLAG_SMALL_REFUSED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- the chunk's groups --------------------------------------------------------------------------------------------------
  // histograms x dictionaries: thread = value id
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) {
    if (c >= plan.n_narrow || tid >= 256) continue;
    const uint32_t narrow_type_c = segment_again(c).data_type;
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) {
      uint32_t count = 0;
      if (plan.joint) {   // the column's histogram = the pair histogram summed over the other column's value ids (its NULL id included)
        const uint32_t second_domain = narrow_size[1] + 1;
        if (c == 0) { for (uint32_t other = 0; other < second_domain && tid < narrow_size[0]; ++other) count += s_hist[(tid * second_domain + other) * SD_DENSE + k]; }
        else { for (uint32_t other = 0; other <= narrow_size[0] && tid < narrow_size[1]; ++other) count += s_hist[(other * second_domain + tid) * SD_DENSE + k]; }
      } else {
#pragma unroll
        for (uint32_t copy_index = 0; copy_index < SD_COPIES; ++copy_index) count += s_hist[((copy_index * SD_NARROW + c) * 256 + tid) * SD_DENSE + k];
      }
      if (tid >= narrow_size[c]) count = 0;
      const bool integer = sd_is_integer(narrow_type_c), track = (plan.extremes >> c) & 1u;
      const uint64_t entry = s_dict[c][tid];
      const uint64_t weighted = integer ? static_cast<uint64_t>(static_cast<int64_t>(entry) * static_cast<int64_t>(count))
                                        : static_cast<uint64_t>(__double_as_longlong(static_cast<double>(count) * __longlong_as_double(static_cast<long long>(entry))));
      const uint64_t sum = sd_wave_sum(weighted, integer);
      const uint32_t total = wave_reduce_u32_to_lane63(count, 0u, false, false);
      uint32_t low = 0xFFFFFFFFu, high = 0;
      if (track) {
        low = wave_reduce_u32_to_lane63(count ? tid : 0xFFFFFFFFu, 0xFFFFFFFFu, true, false);
        high = wave_reduce_u32_to_lane63(count ? tid : 0u, 0u, false, true);
      }
      if (lane == 63 && total) {
        sd_add(&s_sum[k][c], sum, integer);
        atomicAdd(&s_nonnull[k][c], total);
        if (track) { atomicMin(&s_min_id[k][c], low); atomicMax(&s_max_id[k][c], high); }
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
    const uint32_t function = a.aggregates[g].function, inputs = s_nonnull[tid][c];
    if (inputs == 0) continue;
    const DevSegment seg = plan.column[c][chunk];   // (read again: a register array indexed by c would live in scratch memory)
    uint64_t bits = s_sum[tid][c];
    if (function == HY_AGG_MIN || function == HY_AGG_MAX) bits = contribution_from(a.aggregates[g], sd_dictionary_bits(seg.aux, seg.data_type, function == HY_AGG_MIN ? s_min_id[tid][c] : s_max_id[tid][c]));
    else if (function == HY_AGG_AVG && sd_is_integer(seg.data_type)) bits = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<int64_t>(bits))));   // (AVG adds doubles)
    merge_global(a, gslot, g, bits, inputs);
  }
}
