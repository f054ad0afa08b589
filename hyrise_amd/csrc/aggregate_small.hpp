// aggregate_small.hpp -- AggregateHash for the TPC-H Q1 shape (included by aggregate.hip, inside namespace hy).
//
// What it replaces (reference, CPU): the row loop of AggregateHash::_aggregate for a handful of groups
// (operators/aggregate_hash.cpp:317-403 get_or_add_result, :605-655 _aggregate_segment, :1016-1176) -- config 4 of BASELINE.json:
// GROUP BY l_returnflag, l_linestatus (dictionary segments, a few distinct values) with SUM / AVG / COUNT over DictionarySegment<float>
// columns.  aggregate_rows handles every encoding x type x function in one 150 KB body at 167 registers and ran this shape at 11 %
// of the HBM roofline; these kernels take the shape and nothing else:
//   * at most two GROUP BY columns, each a dictionary segment with 1-byte value ids in every chunk, the product of their (dictionary
//     size + 1) at most 16: a row's group is the mixed-radix CODE of its value ids; the first four codes a chunk meets are its DENSE
//     groups (Q1 has four),
//   * every aggregate is SUM / AVG / COUNT / MIN / MAX over a dictionary-encoded int / long / float / double column with 1- or 2-byte
//     value ids (or COUNT(*)): int and long columns sum in int64 (exact, AVG's double is that sum converted), MIN / MAX are the
//     smallest / largest value id a group COUNTED (dictionaries are sorted) looked up once per chunk,
//   * chunks of at most 65536 rows (Hyrise's are 65535).
// Two launches (round 6; rounds 3-5 did all of it in ONE kernel of one 1024-thread workgroup per chunk and CU -- 128 KB of counters --
// whose phases, zero / count / weigh / merge, ran one after the other on every CU: 0.29 ms for three rounds, 0.25 of the roofline):
//   sd_groups  one 256-thread workgroup per chunk, ~22 KB of LDS, four or more per CU: the GROUP BY ids and the 1-byte columns.  A row's
//              dense group (4 bits) is written to a NIBBLE stream -- half a byte per row -- for the second kernel; the 1-byte columns'
//              rows are COUNTED per (value id, dense group) in an LDS histogram -- one LDS atomic per row, no dictionary gather at all --
//              and the counts are weighted with the dictionary once per chunk (the double sums are exact for these columns' products
//              count x value in any order); the chunk's groups enter the global table and their slots are left behind per chunk.
//   sd_wide    the 2-byte column (l_extendedprice: 240 KB of dictionary per chunk, nearly a value per row).  One 512-thread workgroup per
//              (chunk, HALF of the value-id range): 64 KB of 4-bit counters -- one per (value id, dense group) -- so two workgroups share
//              a CU and one's counting overlaps the other's weighing; it reads the nibbles and the ids (the second half's reads come from
//              the XCD's L2: the two halves of a chunk are dispatched side by side on one XCD), counts the rows of its half with one LDS
//              atomic each and weights the counters with its half of the dictionary, read once, coalesced.
// Rows of a fifth, sixth ... group of a chunk take LDS atomics on shared cells in sd_groups (every column).  The chunk's groups are
// merged into the global table like aggregate_rows' (global_slot / merge_global): result order, representative rows and values are those
// of the generic kernel.
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
  uint32_t debug;                                 // HY_AGG_SMALL_DEBUG (timing experiments, wrong results): 1 no histograms, 2 no 2-byte column, 8 no dense lookup, 16 no nibble stream, 32 no rows / first rows
  // between the two launches (device memory of the call): a row's dense group, 4 bits each, chunk c's at nibbles + c * SD_NIBBLE_BYTES (0xF: a
  // row that does not exist or whose group has no dense index below 16); the global-table slots of chunk c's four dense groups (0xFFFFFFFF: none)
  uint8_t* nibbles;
  uint32_t* chunk_slots;
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

constexpr uint32_t SD_MAX_CHUNK_ROWS = 65536;               // a chunk is one pass of either kernel (Hyrise's chunks have at most 65535 rows)
constexpr uint32_t SD_NIBBLE_BYTES = SD_MAX_CHUNK_ROWS / 2; // a chunk's region of the nibble stream
#ifndef HY_SD_THREADS
#define HY_SD_THREADS 256
#endif
#ifndef HY_SD_RING
#define HY_SD_RING 3
#endif
constexpr uint32_t SD_THREADS = HY_SD_THREADS;              // sd_groups: one workgroup per chunk, steps of sixteen rows per lane
constexpr uint32_t SD_RING = HY_SD_RING;                    // register buffers of a lane's loads: steps requested ahead + 1
constexpr uint32_t SD_COPIES = 2;                           // copies of the narrow columns' histograms (even / odd lanes)
constexpr uint32_t SD_HIST_WORDS = SD_COPIES * SD_NARROW * 256 * SD_DENSE;   // (= SD_JOINT_CELLS * SD_DENSE: the pair histogram takes the same words)
static_assert(SD_JOINT_CELLS * SD_DENSE == SD_HIST_WORDS, "the pair histogram lives in the two single histograms' words");

// What a lane asks for per step: sixteen consecutive rows of every column sd_groups reads (16 bytes of 1-byte ids each).
struct SdStep {
  u32x4 key[SD_KEYS];
  u32x4 narrow[SD_NARROW];
};
__device__ __forceinline__ uint32_t sd_word(const u32x4& v, uint32_t w) { return w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w; }
constexpr uint32_t SD_HIST_GROUPS = SD_DENSE + 1;   // a histogram row: the four dense groups and a spare cell (rows of other groups, rows that do not exist)
constexpr uint32_t SD_HIST_CELLS = SD_HIST_WORDS / SD_DENSE * SD_HIST_GROUPS;

// ---- launch 1: GROUP BY ids and 1-byte columns ---------------------------------------------------------------------------------------
// The row loop works on FOUR rows per instruction wherever the rows' bytes allow it: a row's code (mixed radix of its GROUP BY ids) is one
// multiply-add on the dword that holds four rows' ids; its dense index comes out of a 16-byte table (code -> dense index, 0xFF: not met yet)
// with two byte permutes; the groups a lane meets in a step are one more permute (dense index -> one-hot).  No branch per row: rows that do
// not exist and rows of a fifth, sixth ... group count in a spare cell of their histogram row.  Loads are plain loads issued a step ahead
// (rounds 3-5 declared them volatile "so that the compiler does not sink them": it then waited for every one of them where it was issued --
// four dependent round trips per step, most of the old kernel's time).
__global__ __launch_bounds__(SD_THREADS, SD_THREADS == 256 ? 4 : 7) void sd_groups(AggArgs a, SmallDomainPlan plan, uint32_t n_chunks) {
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[SD_HIST_CELLS];   // [copy][narrow column][value id][dense group | spare] rows, or [pair of value ids][dense group | spare]
  __shared__ uint64_t s_dict[SD_NARROW][256];                  // narrow columns' dictionaries (sd_dictionary_bits)
  __shared__ uint32_t s_dense_of_code[SD_CODES];               // 0xFF unassigned, 0xFE being assigned, else the dense index (may be >= SD_DENSE: a shared-cell group)
  __shared__ uint32_t s_code_of_dense[SD_CODES];
  __shared__ uint32_t s_n_dense;
  __shared__ __attribute__((aligned(16))) uint32_t s_lut[4];   // sixteen bytes: dense index of code c, 0xFF until a row with that code has been met
  __shared__ uint64_t s_sum[SD_CODES][SD_COLUMNS];             // per dense index (all of them) and column: a double's bits, an int64 for int / long columns
  __shared__ uint32_t s_min_id[SD_CODES][SD_COLUMNS], s_max_id[SD_CODES][SD_COLUMNS];   // smallest / largest value id counted (plan.extremes)
  __shared__ uint32_t s_nonnull[SD_CODES][SD_COLUMNS];
  __shared__ uint32_t s_rows[SD_CODES], s_last[SD_CODES];
  __shared__ __attribute__((aligned(16))) uint32_t s_first[SD_CODES];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const uint64_t chunk_base = a.row_base[chunk];
  if (plan.chunk_slots && tid < SD_DENSE) plan.chunk_slots[chunk * SD_DENSE + tid] = 0xFFFFFFFFu;   // (until the merge below says otherwise)

  // ---- descriptors ---------------------------------------------------------------------------------------------------------
  const char* key_data[SD_KEYS];
  uint32_t key_stride[SD_KEYS], key_size[SD_KEYS];
  uint32_t rows_in_chunk = 0, product = 1;
#pragma unroll
  for (uint32_t g = 0; g < SD_KEYS; ++g) {
    key_data[g] = nullptr;
    key_stride[g] = 0;
    key_size[g] = 0;
    if (g < a.n_groupby) {
      const DevSegment seg = a.groupby[g].segments[chunk];
      key_data[g] = static_cast<const char*>(seg.data);
      key_size[g] = seg.aux_size;
      key_stride[g] = product;
      product *= seg.aux_size + 1;   // + 1: the NULL value id
      rows_in_chunk = seg.size;
    }
  }
  // What the row loop reads of the input columns stays in registers: the value ids and the dictionaries' sizes.  Dictionaries and types
  // are read from the descriptors again where the counts are weighted.
  const char* narrow_data[SD_NARROW];
  uint32_t narrow_size[SD_NARROW];
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) { narrow_data[c] = nullptr; narrow_size[c] = 0; }
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW + SD_WIDE; ++c) {   // (static indices: an array indexed at run time would live in scratch memory)
    if (c >= plan.n_columns) continue;
    const DevSegment seg = plan.column[c][chunk];
    if (a.n_groupby == 0) rows_in_chunk = seg.size;
    if (c < SD_NARROW && c < plan.n_narrow) { narrow_data[c] = static_cast<const char*>(seg.data); narrow_size[c] = seg.aux_size; }
  }
  auto segment_again = [&](uint32_t c) {
    __asm__ volatile("" ::: "memory");   // (a load of its own: not the values of the loads above kept in registers)
    return plan.column[c][chunk];
  };

  // The loads of a step.  ALWAYS the same four loads, whatever the plan (a column that is not there reads a column that is, and its bytes
  // are not used): the compiler counts outstanding loads per control-flow path, and where a path may have issued fewer of them it waits
  // for all -- `s_waitcnt vmcnt(0)` in front of every step, with the next steps' loads just issued (seen in the disassembly).
  const char* some_column = key_data[0] ? key_data[0] : narrow_data[0] ? narrow_data[0] : static_cast<const char*>(plan.column[0][chunk].data);
#pragma unroll
  for (uint32_t g = 0; g < SD_KEYS; ++g) if (!key_data[g]) key_data[g] = some_column;
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) if (!narrow_data[c]) narrow_data[c] = some_column;
  auto load_step = [&](uint32_t step, SdStep& s) {
    const uint32_t first = (step * SD_THREADS + tid) * SD_ROWS;
    const uint32_t row = first < rows_in_chunk ? first : 0;   // (a lane without rows reads the chunk's first ids; a load that starts at an existing row ends inside the padding every uploaded buffer has)
#pragma unroll
    for (uint32_t g = 0; g < SD_KEYS; ++g) s.key[g] = *(const global_u32x4*)(key_data[g] + row);
#pragma unroll
    for (uint32_t c = 0; c < SD_NARROW; ++c) s.narrow[c] = *(const global_u32x4*)(narrow_data[c] + row);
  };
  // Two steps ahead, in THREE FIXED register buffers (the step loop is unrolled by three): a buffer is loaded again right after its step
  // has been worked on.  Rotating two buffers with assignments (current = next; next = load(...)) makes the compiler wait for the load just
  // issued at the end of every step -- the copy reads its registers -- which exposed a full memory round trip per step: 16 x 4 us per
  // chunk, most of this kernel's time until that was seen in the disassembly (and most of rounds 3-5's kernel's).
  SdStep ring[SD_RING];
#pragma unroll
  for (uint32_t r = 0; r < SD_RING; ++r) load_step(r, ring[r]);   // (steps behind the chunk's last read its first rows)

  for (uint32_t i = tid; i < SD_HIST_CELLS; i += SD_THREADS) s_hist[i] = 0;
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) {
    if (c >= plan.n_narrow || tid >= 256) continue;   // (a narrow column's dictionary has at most 255 entries: thread = value id)
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
  if (tid == 0) s_n_dense = 0;
  if (tid < 4) s_lut[tid] = 0xFFFFFFFFu;
  __syncthreads();   // (the tables are set up)

  const uint32_t copy = lane & (SD_COPIES - 1);
  uint8_t* chunk_nibbles = plan.nibbles ? plan.nibbles + size_t{chunk} * SD_NIBBLE_BYTES : nullptr;
  const uint32_t stride1 = key_stride[1], key0_mask = a.n_groupby ? 0xFFFFFFFFu : 0u;   // (the first GROUP BY column's stride is 1; a column that is not there: stride 0)
  const bool track_last = a.n_groupby == 1 && a.groupby[0].data_type == HY_TYPE_INT;   // (the immediate-key shortcut names a group's LAST row, aggregate_hash.cpp:388-401)
  u32x4 lut = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // this lane's copy of the table code -> dense index
  uint32_t rows_of[SD_DENSE] = {0, 0, 0, 0};   // rows of the dense groups this lane has met
  uint32_t first_of[SD_DENSE] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // ... and the first of them
  uint32_t seen = 0;                            // bit k: this lane has met dense group k
  const uint32_t n_steps = (rows_in_chunk + SD_THREADS * SD_ROWS - 1) / (SD_THREADS * SD_ROWS);
  // ---- the rows: groups, bookkeeping, histograms of the 1-byte columns, the nibble stream ---------------------------------------------
  auto work_on = [&](uint32_t step, SdStep& current) __attribute__((always_inline)) {
    const uint32_t first = (step * SD_THREADS + tid) * SD_ROWS;
    // which of the lane's sixteen rows exist: all of them, except in the chunk's last rows
    uint32_t exists[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (__any(first + SD_ROWS > rows_in_chunk)) {
      const uint32_t n = first < rows_in_chunk ? (rows_in_chunk - first < SD_ROWS ? rows_in_chunk - first : SD_ROWS) : 0u;
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) exists[w] = n >= 4 * w + 4 ? 0xFFFFFFFFu : n <= 4 * w ? 0u : (1u << (8 * (n - 4 * w))) - 1u;
    }
    // (the bytes behind a chunk's last row are padding: as value ids they could name a histogram row that is not there)
    if (__any(first + SD_ROWS > rows_in_chunk)) {
#pragma unroll
      for (uint32_t c = 0; c < SD_NARROW; ++c) {
        current.narrow[c].x &= exists[0]; current.narrow[c].y &= exists[1]; current.narrow[c].z &= exists[2]; current.narrow[c].w &= exists[3];
      }
    }
    // codes of four rows per dword: id of the first column + id of the second x its stride (no byte overflows: codes are below sixteen)
    uint32_t code4[4];
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) code4[w] = (sd_word(current.key[0], w) & key0_mask) + sd_word(current.key[1], w) * stride1;
    // dense indices: the table's sixteen bytes, two permutes per dword (codes 0 .. 7 | 8 .. 15) merged by the codes' bit 3
    // The table is kept in REGISTERS and read from LDS again only when a row's code has no entry in the copy (the chunk's first steps; a code
    // another wave entered later): an LDS read per step waits for the wave's sixteen histogram atomics of the step before to drain -- a
    // quarter of this kernel's time when every step did it.
    uint32_t dense4[4];
    if (plan.debug & 8) {
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) dense4[w] = code4[w] & 0x03030303u;
    } else
    for (bool fresh_table = false;; fresh_table = true) {
      if (fresh_table) {
        // (relaxed atomic loads, not volatile ones: a volatile access makes the compiler wait for EVERY outstanding load -- the next step's, just issued)
        lut.x = __hip_atomic_load(&s_lut[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lut.y = __hip_atomic_load(&s_lut[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lut.z = __hip_atomic_load(&s_lut[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lut.w = __hip_atomic_load(&s_lut[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      uint32_t unassigned = 0;
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t select = code4[w] & 0x07070707u;
        const uint32_t low = __builtin_amdgcn_perm(lut.y, lut.x, select), high = __builtin_amdgcn_perm(lut.w, lut.z, select);
        const uint32_t top = (code4[w] >> 3) & 0x01010101u, mask = (top << 8) - top;   // 0xFF in the bytes whose code is 8 .. 15
        dense4[w] = (high & mask) | (low & ~mask);
        unassigned |= dense4[w] & exists[w];
      }
      if (!__any((unassigned & 0x80808080u) != 0)) break;
      if (!fresh_table) continue;   // (the copy is behind: look at the table itself before claiming anything)
      // a code the chunk has not met before (its first rows): a lane claims the code's entry (lanes of a wave run in lockstep: the claimant
      // never waits, the others look again), takes the next dense index and publishes it with ONE atomic on the table's word
#pragma unroll 1
      for (uint32_t j = 0; j < SD_ROWS; ++j) {
        const uint32_t w = j >> 2, shift = 8 * (j & 3);
        const uint32_t d = (w == 0 ? dense4[0] : w == 1 ? dense4[1] : w == 2 ? dense4[2] : dense4[3]) >> shift & 0xFFu;
        const uint32_t here = (w == 0 ? exists[0] : w == 1 ? exists[1] : w == 2 ? exists[2] : exists[3]) >> shift & 0xFFu;
        if (d != 0xFFu || !here) continue;
        const uint32_t code = (w == 0 ? code4[0] : w == 1 ? code4[1] : w == 2 ? code4[2] : code4[3]) >> shift & 0xFu;
        if (atomicCAS(&s_dense_of_code[code], 0xFFu, 0xFEu) == 0xFFu) {
          const uint32_t assigned = atomicAdd(&s_n_dense, 1u);
          s_code_of_dense[assigned] = code;
          __hip_atomic_store(&s_dense_of_code[code], assigned, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_and(&s_lut[code >> 2], ~(0xFFu << (8 * (code & 3))) | assigned << (8 * (code & 3)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) dense4[w] |= ~exists[w];   // (rows that do not exist: 0xFF, no group)
    // the second launch reads the rows' groups here: a nibble per row, eight bytes per lane and step, the lanes' side by side
    if (chunk_nibbles && first < rows_in_chunk && !(plan.debug & 16)) {
      uint32_t packed[4];
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) {
        uint32_t x = dense4[w] & 0x0F0F0F0Fu;
        x = (x | x >> 4) & 0x00FF00FFu;
        packed[w] = (x | x >> 8) & 0xFFFFu;
      }
      const uint64_t nibbles = static_cast<uint64_t>(packed[0] | packed[1] << 16) | static_cast<uint64_t>(packed[2] | packed[3] << 16) << 32;
      __builtin_nontemporal_store(nibbles, reinterpret_cast<uint64_t*>(chunk_nibbles + first / 2));
    }
    // one-hot of the dense groups (bit k of a row's byte: the row is of dense group k; nothing for other groups and rows that do not exist)
    if (!(plan.debug & 32)) {
    uint32_t hot[4];
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) hot[w] = __builtin_amdgcn_perm(0u, 0x08040201u, (dense4[w] & 0x07070707u) | ((dense4[w] >> 1) & 0x04040404u));
    {   // rows per dense group: two rows per byte, a mask and a population count per group and half
      const uint32_t even = hot[0] | hot[1] << 4, odd = hot[2] | hot[3] << 4;
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) rows_of[k] += __popc(even & (0x11111111u << k)) + __popc(odd & (0x11111111u << k));
    }
    uint32_t present = hot[0] | hot[1] | hot[2] | hot[3];
    present |= present >> 16;
    present = (present | present >> 8) & 0xFu;
    // first row of a group, per lane, in registers (they meet in LDS when the chunk is done): where a lane meets a group for the first
    // time -- a rare group, Q1's N / F, is met for the first time by some lane of a wave in nearly every step -- the position of the group's
    // first byte among the sixteen.  No LDS access here: one would wait for the step's atomics.
    const uint32_t fresh = present & ~seen;
    seen |= present;
    if (__any(fresh != 0)) {
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        // bit 8 j + k of the dwords: row 4 w + j is of group k; the first such row
        const uint32_t m0 = hot[0] & (0x01010101u << k), m1 = hot[1] & (0x01010101u << k), m2 = hot[2] & (0x01010101u << k), m3 = hot[3] & (0x01010101u << k);
        const uint32_t word = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u, bits = m0 ? m0 : m1 ? m1 : m2 ? m2 : m3;
        const uint32_t position = 4 * word + ((__ffs(bits) - 1) >> 3);
        first_of[k] = ((fresh >> k) & 1u) ? first + position : first_of[k];
      }
    }
    if (track_last && present) {   // (the immediate-key shortcut names a group's last row: only plans with one int32 GROUP BY column come here)
#pragma unroll 1
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        if (!((present >> k) & 1u)) continue;
        uint32_t highest = 0;
#pragma unroll 1
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t w = j >> 2;
          const uint32_t byte = (w == 0 ? hot[0] : w == 1 ? hot[1] : w == 2 ? hot[2] : hot[3]) >> (8 * (j & 3)) & 0xFFu;
          if ((byte >> k) & 1u) highest = first + j;
        }
        atomicMax(&s_last[k], highest);
      }
    }
    }
    // a fifth, sixth ... group of this chunk: shared LDS cells, row by row, every column (rare)
    const uint32_t beyond = (dense4[0] | dense4[1] | dense4[2] | dense4[3]) & 0x0C0C0C0Cu;
    if (__any(beyond != 0)) {
#pragma unroll 1
      for (uint32_t j = 0; j < SD_ROWS; ++j) {
        const uint32_t w = j >> 2;
        const uint32_t d = (w == 0 ? dense4[0] : w == 1 ? dense4[1] : w == 2 ? dense4[2] : dense4[3]) >> (8 * (j & 3)) & 0xFFu;
        if (d < SD_DENSE || d == 0xFFu) continue;
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
    // 1-byte value ids: count the row in the histogram row of its value id(s), in the cell of its dense group -- the spare cell for rows
    // of other groups and rows that do not exist; NULL ids (the dictionary's size) are counted like the others and left out when the
    // counts are weighted.  Cells of one value id are neighbours (a column with eleven values, l_discount, would otherwise meet in eleven
    // of the LDS's banks).
    if (plan.joint && !(plan.debug & 1)) {
      // Two 1-byte columns (l_quantity x l_discount: 51 x 12 pairs): the row counts ONCE, in the cell of its pair of value ids -- one LDS
      // atomic instead of two; the pair counts are summed into each column's histogram when the chunk is done.
      const uint32_t second_domain = narrow_size[1] + 1;
#pragma unroll
      for (uint32_t j = 0; j < SD_ROWS; ++j) {
        const uint32_t w = j >> 2, shift = 8 * (j & 3);
        const uint32_t d = (dense4[w] >> shift) & 0xFFu;
        const uint32_t id0 = (sd_word(current.narrow[0], w) >> shift) & 0xFFu, id1 = (sd_word(current.narrow[1], w) >> shift) & 0xFFu;
        const uint32_t cell = id0 * second_domain + id1;
        atomicAdd(&s_hist[cell * SD_HIST_GROUPS + (d < SD_DENSE ? d : SD_DENSE)], 1u);
      }
    } else {
#pragma unroll
      for (uint32_t c = 0; c < SD_NARROW; ++c) {
        if (c >= plan.n_narrow || (plan.debug & 1)) continue;
        uint32_t* cells = s_hist + (copy * SD_NARROW + c) * 256 * SD_HIST_GROUPS;
#pragma unroll
        for (uint32_t j = 0; j < SD_ROWS; ++j) {
          const uint32_t w = j >> 2, shift = 8 * (j & 3);
          const uint32_t d = (dense4[w] >> shift) & 0xFFu;
          const uint32_t id = (sd_word(current.narrow[c], w) >> shift) & 0xFFu;
          atomicAdd(&cells[id * SD_HIST_GROUPS + (d < SD_DENSE ? d : SD_DENSE)], 1u);
        }
      }
    }
  };
#pragma unroll 1
  for (uint32_t step = 0; step < n_steps; step += SD_RING) {
#pragma unroll
    for (uint32_t r = 0; r < SD_RING; ++r) {
      if (step + r >= n_steps) break;
      work_on(step + r, ring[r]);
      load_step(step + r + SD_RING, ring[r]);   // (unconditional, see load_step)
    }
  }
  // the chunk's dense groups: rows -> the LDS cells (first and last rows are there already)
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) {
    const uint32_t rows = wave_reduce_u32_to_lane63(rows_of[k], 0u, false, false);
    const uint32_t lowest = wave_reduce_u32_to_lane63(first_of[k], 0xFFFFFFFFu, true, false);
    if (lane == 63 && rows) { atomicAdd(&s_rows[k], rows); atomicMin(&s_first[k], lowest); }
  }
  __syncthreads();

  // ---- the chunk's groups --------------------------------------------------------------------------------------------------
  // histograms x dictionaries: thread = value id
#pragma unroll
  for (uint32_t c = 0; c < SD_NARROW; ++c) {
    if (c >= plan.n_narrow || tid >= 256) continue;   // (whole waves)
    const uint32_t narrow_type_c = segment_again(c).data_type;
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) {
      uint32_t count = 0;
      if (plan.joint) {   // the column's histogram = the pair histogram summed over the other column's value ids (its NULL id included)
        const uint32_t second_domain = narrow_size[1] + 1;
        if (c == 0) { for (uint32_t other = 0; other < second_domain && tid < narrow_size[0]; ++other) count += s_hist[(tid * second_domain + other) * SD_HIST_GROUPS + k]; }
        else { for (uint32_t other = 0; other <= narrow_size[0] && tid < narrow_size[1]; ++other) count += s_hist[(other * second_domain + tid) * SD_HIST_GROUPS + k]; }
      } else {
#pragma unroll
        for (uint32_t copy_index = 0; copy_index < SD_COPIES; ++copy_index) count += s_hist[((copy_index * SD_NARROW + c) * 256 + tid) * SD_HIST_GROUPS + k];
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
  if (plan.chunk_slots && tid < SD_DENSE) plan.chunk_slots[chunk * SD_DENSE + tid] = gslot;   // (sd_wide merges the 2-byte column's sums there)
  atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(chunk_base + s_first[tid]));
  atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(chunk_base + (track_last || tid >= SD_DENSE ? s_last[tid] : s_first[tid])));
  for (uint32_t g = 0; g < a.n_aggregates; ++g) {
    const uint32_t c = plan.column_of_aggregate[g];
    if (c == 0xFFFFFFFFu) { merge_global(a, gslot, g, 0, s_rows[tid]); continue; }
    // (a 2-byte column: only what the shared-cell path above summed, for a fifth, sixth ... group; the dense groups' are sd_wide's)
    const uint32_t function = a.aggregates[g].function, inputs = s_nonnull[tid][c];
    if (inputs == 0) continue;
    const DevSegment seg = plan.column[c][chunk];   // (read again: a register array indexed by c would live in scratch memory)
    uint64_t bits = s_sum[tid][c];
    if (function == HY_AGG_MIN || function == HY_AGG_MAX) bits = contribution_from(a.aggregates[g], sd_dictionary_bits(seg.aux, seg.data_type, function == HY_AGG_MIN ? s_min_id[tid][c] : s_max_id[tid][c]));
    else if (function == HY_AGG_AVG && sd_is_integer(seg.data_type)) bits = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<int64_t>(bits))));   // (AVG adds doubles)
    merge_global(a, gslot, g, bits, inputs);
  }
}

// ---- launch 2: the 2-byte column ---------------------------------------------------------------------------------------------------------
// History (l_extendedprice, 240 KB of dictionary per chunk -- more than a CU's L1, and 60 M gathers that each pull a 128-byte line out of the
// L2 for four bytes were 700 of the first version's 980 us and are what bounds aggregate_rows): gathers from dictionary windows staged in
// LDS, 550 us; rows COUNTED per (dense group, value id) in 16-bit LDS counters, eight windows of 8192 ids, 520 us; ONE window of 4-bit
// counters -- 128 KB of LDS for 65536 ids, one 1024-thread workgroup per chunk and CU -- 290 us (rounds 3-5).  Now the value-id range in
// SD_WIDE_PARTS parts of 128 KB / SD_WIDE_PARTS of counters: a row costs one LDS atomic in the part its id falls in, the ids are read once
// from HBM and again from the L2, the dictionary is read once, coalesced, when the counts are weighted.  A counter that meets a sixteenth
// row carries into its neighbour; every carry lowers the sum of all counters, which is compared with the rows counted in registers: a part
// that fails the comparison (sixteen rows of one group with one price in 65535 rows) raises FLAG_SMALL_REFUSED and the host runs
// aggregate_rows instead.
#ifndef HY_SD_WIDE_PARTS
#define HY_SD_WIDE_PARTS 1
#endif
constexpr uint32_t SD_WIDE_PARTS = HY_SD_WIDE_PARTS;        // 1: the whole value-id range, 128 KB of counters, one 1024-thread workgroup per chunk and CU; 2 | 4: see above
constexpr uint32_t SD_WIDE_IDS = 65536 / SD_WIDE_PARTS;     // value ids of a part
constexpr uint32_t SD_WIDE_BYTES = SD_WIDE_IDS * 2;         // one 16-bit cell per value id: four 4-bit counters, one per dense group
constexpr uint32_t SD_WIDE_THREADS = SD_WIDE_PARTS == 1 ? 1024 : 512;
constexpr uint32_t SD_WIDE_RESIDENT = SD_WIDE_PARTS == 1 ? 1 : SD_WIDE_PARTS == 2 ? 2 : 4;   // workgroups per CU the LDS admits (and the register budget is cut for)
constexpr uint32_t SD_WIDE_TAIL_BYTES = 128;                // behind the counters: the part's sums, counts and extremes
__host__ __device__ constexpr size_t sd_wide_lds_bytes() { return SD_WIDE_BYTES + SD_WIDE_TAIL_BYTES; }

// The counters of a part x its dictionary entries (Stored: the dictionary's element type).  Thread t takes the PAIRS of value ids 2 t, 2 t + 1,
// then 2 (t + SD_WIDE_THREADS) ...: a pair's two counters are one LDS word, its two dictionary entries one load; BATCH pairs are requested
// at once -- they come from HBM, coalesced, and one request after the other was a sixth of the old kernel -- and the first batch is
// requested BEFORE the barrier that ends the counting (the dictionary does not depend on it).  Out, per dense group: the sum (a double's
// bits; an int64 for int / long) and the inputs counted (the counters' nibbles summed four fields per add, widened every sixteen words).
template <typename Stored>
__device__ __forceinline__ void sd_weigh_wide(const void* dictionary, uint32_t first_id, uint32_t end_id, const uint32_t* s_wide, uint32_t tid, uint64_t (&sums)[SD_DENSE],
                                              uint32_t (&counted)[SD_DENSE]) {
  typedef const __attribute__((address_space(1))) Stored global_stored;
  constexpr bool INTEGER = std::is_integral<Stored>::value;
  typedef typename std::conditional<INTEGER, int64_t, double>::type Sum;
  constexpr uint32_t BATCH = sizeof(Stored) == 4 ? 16 : 8;
  constexpr uint32_t PAIRS = SD_WIDE_IDS / 2 / SD_WIDE_THREADS;   // pairs of a thread
  static_assert(PAIRS % BATCH == 0 && BATCH <= 16, "whole batches; the packed count fields take sixteen words of counters");
  Sum acc[SD_DENSE];
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) { acc[k] = 0; counted[k] = 0; }
  const uint32_t last_id = end_id - 1;   // (end_id > first_id: the caller returned otherwise)
  bool counting_done = false;
#pragma unroll 1
  for (uint32_t batch = 0; batch < PAIRS; batch += BATCH) {
    if (first_id + 2 * batch * SD_WIDE_THREADS >= end_id) break;
    Stored value[BATCH][2];
    if (first_id + 2 * (batch + BATCH) * SD_WIDE_THREADS <= end_id) {   // (uniform) the whole batch lies inside the dictionary: a pair is one load
      typedef Stored StoredPair __attribute__((ext_vector_type(2)));
      typedef const __attribute__((address_space(1))) StoredPair global_pair;
#pragma unroll
      for (uint32_t n = 0; n < BATCH; ++n) {
        const StoredPair pair = ((global_pair*)dictionary)[first_id / 2 + (batch + n) * SD_WIDE_THREADS + tid];
        value[n][0] = pair.x;
        value[n][1] = pair.y;
      }
    } else {
#pragma unroll
      for (uint32_t n = 0; n < BATCH; ++n) {
        const uint32_t i = first_id + 2 * ((batch + n) * SD_WIDE_THREADS + tid);
        // (ids behind the dictionary's end read its last entry: their counters are zero)
        value[n][0] = ((global_stored*)dictionary)[i < last_id ? i : last_id];
        value[n][1] = ((global_stored*)dictionary)[i + 1 < last_id ? i + 1 : last_id];
      }
    }
    if (!counting_done) { __syncthreads(); counting_done = true; }
    uint32_t even_fields = 0, odd_fields = 0;   // bytes: id 0 group 0 | id 0 group 2 | id 1 group 0 | id 1 group 2, and groups 1 | 3
#pragma unroll
    for (uint32_t n = 0; n < BATCH; ++n) {
      const uint32_t word = s_wide[(batch + n) * SD_WIDE_THREADS + tid];
      even_fields += word & 0x0F0F0F0Fu;
      odd_fields += (word >> 4) & 0x0F0F0F0Fu;
#pragma unroll
      for (uint32_t i = 0; i < 2; ++i) {
#pragma unroll
        for (uint32_t k = 0; k < SD_DENSE; ++k) {
          const uint32_t count = (word >> (16 * i + 4 * k)) & 0xFu;
          acc[k] += static_cast<Sum>(count) * static_cast<Sum>(value[n][i]);
        }
      }
    }
    counted[0] += (even_fields & 0xFFu) + ((even_fields >> 16) & 0xFFu);
    counted[2] += ((even_fields >> 8) & 0xFFu) + (even_fields >> 24);
    counted[1] += (odd_fields & 0xFFu) + ((odd_fields >> 16) & 0xFFu);
    counted[3] += ((odd_fields >> 8) & 0xFFu) + (odd_fields >> 24);
  }
  if (!counting_done) __syncthreads();   // (an empty part)
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) {
    if constexpr (INTEGER) sums[k] = static_cast<uint64_t>(acc[k]);
    else sums[k] = static_cast<uint64_t>(__double_as_longlong(acc[k]));
  }
}

__global__ __launch_bounds__(SD_WIDE_THREADS, SD_WIDE_RESIDENT * SD_WIDE_THREADS / 256) void sd_wide(AggArgs a, SmallDomainPlan plan, uint32_t n_chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sd_smem[];
  // (no static LDS in this kernel: the counters start at LDS address 0, a counter's address is its index and nothing is added per row)
  uint32_t* s_wide = reinterpret_cast<uint32_t*>(sd_smem);   // [SD_WIDE_IDS / 2] two 16-bit cells each
  uint64_t* s_sum = reinterpret_cast<uint64_t*>(sd_smem + SD_WIDE_BYTES);   // [SD_DENSE] a double's bits, an int64 for int / long columns
  uint32_t* s_nonnull = reinterpret_cast<uint32_t*>(sd_smem + SD_WIDE_BYTES + 32), *s_min_id = s_nonnull + SD_DENSE, *s_max_id = s_min_id + SD_DENSE;
  uint32_t& s_check = *(s_max_id + SD_DENSE);                // counters summed - rows counted: not zero = a counter overflowed
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  // the parts of a chunk are neighbours in dispatch order on ONE XCD (block b runs on XCD b % 8): the part that comes second finds the
  // chunk's nibbles and ids in that XCD's L2
  const uint32_t group = blockIdx.x / (8 * SD_WIDE_PARTS), within = blockIdx.x % (8 * SD_WIDE_PARTS);
  const uint32_t chunk = group * 8 + within % 8, part = within / 8;
  if (chunk >= n_chunks) return;
  const uint32_t c = plan.n_narrow;   // the column
  const DevSegment wide = plan.column[c][chunk];
  const uint32_t rows_in_chunk = wide.size, wide_size = wide.aux_size;
  const uint32_t first_id = part * SD_WIDE_IDS, end_id = wide_size < first_id + SD_WIDE_IDS ? wide_size : first_id + SD_WIDE_IDS;
  if (first_id >= wide_size) return;   // (no value id of this part exists)
  const uint8_t* chunk_nibbles = plan.nibbles + size_t{chunk} * SD_NIBBLE_BYTES;

  struct Step { u32x4 ids[2]; uint64_t groups; };
  const char* id_data = static_cast<const char*>(wide.data);
  // the loads of a step: plain loads, issued before the step before is worked on (no branch: a lane without rows, or without a second
  // half, reads bytes it has a right to and gets the nibbles of rows that do not exist)
  auto load_step = [&](uint32_t step, Step& s) {
    typedef const __attribute__((address_space(1))) uint64_t global_u64;
    const uint32_t first = (step * SD_WIDE_THREADS + tid) * SD_ROWS;
    const uint32_t row = first < rows_in_chunk ? first : 0;
    const char* at = id_data + size_t{row} * 2;
    s.ids[0] = *(const global_u32x4*)at;
    s.ids[1] = *(const global_u32x4*)(row + 8 < rows_in_chunk ? at + 16 : at);   // (a load starts at an existing row: it ends inside the padding every uploaded buffer has)
    s.groups = *(global_u64*)(chunk_nibbles + row / 2);   // (a lane without rows: masked where the word is used -- not here, which would wait for the load)
  };
  const uint32_t n_steps = (rows_in_chunk + SD_WIDE_THREADS * SD_ROWS - 1) / (SD_WIDE_THREADS * SD_ROWS);
  Step ring[3];   // (two steps ahead in three fixed buffers, every load unconditional: like sd_groups)
  load_step(0, ring[0]);
  load_step(1, ring[1]);
  load_step(2, ring[2]);
  for (uint32_t i = tid; i < SD_WIDE_BYTES / 16; i += SD_WIDE_THREADS) reinterpret_cast<u32x4*>(sd_smem)[i] = u32x4{0, 0, 0, 0};
  if (tid < SD_DENSE) { s_sum[tid] = 0; s_nonnull[tid] = 0; s_min_id[tid] = 0xFFFFFFFFu; s_max_id[tid] = 0; }
  if (tid == 0) s_check = 0;
  __syncthreads();
  uint32_t wide_rows = 0;   // rows this lane counted
  const uint32_t n_local = end_id - first_id;
  auto work_on = [&](uint32_t step, const Step& current) __attribute__((always_inline)) {
    // One 4-bit counter per (value id, dense group).  NULL ids (the dictionary's size), ids of the other parts and rows without a dense group
    // (nibble 4 .. 15: sd_groups summed them itself, or they do not exist) add ZERO to a word of the part -- no branch per row.
    const bool has_rows = (step * SD_WIDE_THREADS + tid) * SD_ROWS < rows_in_chunk;   // (sd_groups wrote 0xF for the rows of a lane's sixteen that do not exist)
    const uint32_t groups_low = has_rows ? static_cast<uint32_t>(current.groups) : 0xFFFFFFFFu, groups_high = has_rows ? static_cast<uint32_t>(current.groups >> 32) : 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t j = 0; j < SD_ROWS; ++j) {
      const uint32_t nibble = ((j < 8 ? groups_low : groups_high) >> (4 * (j & 7))) & 0xFu;
      const uint32_t id = sd_id(current.ids, 2u, j);
      const uint32_t local = id - first_id;   // (an id below the part wraps around: not below n_local)
      const bool counts = (nibble & 0xCu) == 0 && local < n_local;
      const uint32_t one = counts ? 1u : 0u;
      atomicAdd(&s_wide[(local >> 1) & (SD_WIDE_IDS / 2 - 1)], one << (((id << 4) & 16u) | (nibble << 2 & 12u)));
      wide_rows += one;
    }
  };
#pragma unroll 1
  for (uint32_t step = 0; step < n_steps; step += 3) {
#pragma unroll
    for (uint32_t r = 0; r < 3; ++r) {
      if (step + r >= n_steps) break;
      work_on(step + r, ring[r]);
      load_step(step + r + 3, ring[r]);
    }
  }
  // ---- the counters x the dictionary ----------------------------------------------------------------------------------------------------
  const bool integer = sd_is_integer(wide.data_type), track = (plan.extremes >> c) & 1u;
  uint64_t acc[SD_DENSE];                      // sums: doubles' bits, int64 for int / long columns
  uint32_t counted[SD_DENSE] = {0, 0, 0, 0};   // non-NULL inputs per dense group
  switch (wide.data_type) {   // (each ends the counting with a workgroup barrier)
    case HY_TYPE_INT: sd_weigh_wide<int32_t>(wide.aux, first_id, end_id, s_wide, tid, acc, counted); break;
    case HY_TYPE_LONG: sd_weigh_wide<int64_t>(wide.aux, first_id, end_id, s_wide, tid, acc, counted); break;
    case HY_TYPE_FLOAT: sd_weigh_wide<float>(wide.aux, first_id, end_id, s_wide, tid, acc, counted); break;
    default: sd_weigh_wide<double>(wide.aux, first_id, end_id, s_wide, tid, acc, counted); break;
  }
  // the sums -> LDS cells (a wave reduction each, then one LDS atomic per wave); counters summed against rows counted
  uint32_t all_counted = 0;
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) {
    const uint64_t sum = sd_wave_sum(acc[k], integer);
    const uint32_t inputs = wave_reduce_u32_to_lane63(counted[k], 0u, false, false);
    all_counted += inputs;
    if (lane == 63) {
      sd_add(&s_sum[k], sum, integer);
      atomicAdd(&s_nonnull[k], inputs);
    }
  }
  if (track) {   // MIN / MAX: the smallest and largest value id every dense group counted -- a pass of its own over the counters
    uint32_t low[SD_DENSE], high[SD_DENSE];   // (high: the largest value id + 1, 0 = none)
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) { low[k] = 0xFFFFFFFFu; high[k] = 0; }
#pragma unroll 4
    for (uint32_t local = tid; local < end_id - first_id; local += SD_WIDE_THREADS) {   // (a thread's value ids ascend)
      const uint32_t cell = (s_wide[local >> 1] >> (16 * (local & 1))) & 0xFFFFu;
#pragma unroll
      for (uint32_t k = 0; k < SD_DENSE; ++k) {
        const bool counted_here = ((cell >> (4 * k)) & 0xFu) != 0;
        low[k] = counted_here ? min(low[k], first_id + local) : low[k];
        high[k] = counted_here ? first_id + local + 1 : high[k];
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) {
      const uint32_t lowest = wave_reduce_u32_to_lane63(low[k], 0xFFFFFFFFu, true, false);
      const uint32_t highest = wave_reduce_u32_to_lane63(high[k], 0u, false, true);
      if (lane == 63 && highest) { atomicMin(&s_min_id[k], lowest); atomicMax(&s_max_id[k], highest - 1); }
    }
  }
  const uint32_t expected = wave_reduce_u32_to_lane63(wide_rows, 0u, false, false);
  if (lane == 63 && all_counted != expected) atomicAdd(&s_check, all_counted - expected);
  __syncthreads();
  if (tid == 0 && s_check != 0) __hip_atomic_store(&a.overflow[FLAG_SMALL_REFUSED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // merge into the global table: thread = dense index, the slot sd_groups left behind
  if (tid >= SD_DENSE) return;
  const uint32_t inputs = s_nonnull[tid];
  if (inputs == 0) return;
  if (__hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const uint32_t gslot = plan.chunk_slots[chunk * SD_DENSE + tid];
  if (gslot == 0xFFFFFFFFu) return;   // (sd_groups could not enter the group: it raised the overflow flag)
  for (uint32_t g = 0; g < a.n_aggregates; ++g) {
    if (plan.column_of_aggregate[g] != c) continue;
    const uint32_t function = a.aggregates[g].function;
    uint64_t bits = s_sum[tid];
    if (function == HY_AGG_MIN || function == HY_AGG_MAX) bits = contribution_from(a.aggregates[g], sd_dictionary_bits(wide.aux, wide.data_type, function == HY_AGG_MIN ? s_min_id[tid] : s_max_id[tid]));
    else if (function == HY_AGG_AVG && integer) bits = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<int64_t>(bits))));   // (AVG adds doubles)
    merge_global(a, gslot, g, bits, inputs);
  }
}
