// fused_small.hpp -- hy_scan_project_aggregate for the TPC-H Q1 shape (included by aggregate.hip, inside namespace hy, behind fused_rows).
//
// What it replaces (reference, CPU): TableScan -> Projection -> AggregateHash over one table (operators/table_scan.cpp:171-204,
// operators/projection.cpp:69-218, operators/aggregate_hash.cpp:605-655 and :1016-1176) where
//   * the GROUP BY columns are those of aggregate_small.hpp: at most two dictionary columns with 1-byte value ids whose
//     (dictionary size + 1)s multiply to at most 16 in every chunk -- the first four codes a chunk meets are its DENSE groups,
//   * every filter reads a dictionary column with 1- or 2-byte value ids: a test of the value id against the chunk's job,
//   * every aggregate is SUM / AVG / COUNT over a float expression (+ - * over float columns and literals) or COUNT(*); the
//     expressions read at most four columns, all DictionarySegment<float>, at most one of them with 2-byte value ids.
// fused_rows serves every shape from one body: its survivors' list, the decoded 64-bit stack words and sixteen copies of LDS accumulator
// cells per dense group run Q1 at 3.0 ms (SF10); this kernel at 0.55 ms.  Here a lane takes four consecutive rows per step (4- and 8-byte loads of ids), tests the
// filters on the ids, takes the float values from dictionaries in LDS (the 1-byte columns' dictionaries whole; of the 2-byte column's --
// 240 KB per chunk for l_extendedprice -- the first 38912 entries, the others are gathered through the L2), interprets the postfix
// expressions on float registers (the programs are kernel arguments: five bits per node in scalar registers) and adds the results to
// per-lane double accumulators, one per (accumulator, dense group), selected by the row's group.  One workgroup of 1024 threads per chunk.
// What the kernel does not take -- a fifth group in a chunk, a NULL in an input column, a LIKE filter's value-id set -- raises
// FLAG_SMALL_REFUSED: the host runs fused_rows instead.
// SUM / AVG: double additions in a different order than the reference's row loop -- the stated 1e-9 relative tolerance.
#pragma once

constexpr uint32_t FS_THREADS = 1024;
constexpr uint32_t FS_ROWS = 4;          // consecutive rows of a lane per step (their expressions are evaluated side by side)
constexpr uint32_t FS_STEPS = 16;        // steps per span
constexpr uint32_t FS_SPAN = FS_THREADS * FS_ROWS * FS_STEPS;   // 65536 rows: a Hyrise chunk (at most 65535 rows) is one span; the host sends no larger chunk here
constexpr uint32_t FS_FILTERS = 2;
constexpr uint32_t FS_NARROW = 3;        // columns with 1-byte value ids the expressions read: slots 0 .. 2
constexpr uint32_t FS_COLUMNS = FS_NARROW + 1;   // slot 3: the column with 2-byte value ids
constexpr uint32_t FS_INPUTS = 5;        // accumulators with an expression
constexpr uint32_t FS_WINDOW = 38912;    // entries of the 2-byte column's dictionary in LDS (the workgroup takes all 160 KB of its CU)
constexpr uint32_t FS_LITERALS = 4;      // distinct literals of the expressions
enum : uint32_t { FS_PUSH_COLUMN = 0 /* + slot */, FS_PUSH_LITERAL = 4 /* + index */, FS_ADD = 8, FS_SUB = 9, FS_MUL = 10 };   // a node of a program: five bits

// The expression stack's cell: two rows' floats side by side (packed float instructions), or -- HY_FS_SCALAR_STACK, for timing -- one.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef HY_FS_SCALAR_STACK
typedef float fs_cell;
constexpr int FS_CELLS = static_cast<int>(FS_ROWS);
__device__ __forceinline__ fs_cell fs_cell_of(float v) { return v; }
#define FS_ROW(cells, i) ((cells)[i])
#else
typedef f32x2 fs_cell;
constexpr int FS_CELLS = static_cast<int>(FS_ROWS) / 2;
__device__ __forceinline__ fs_cell fs_cell_of(float v) { return f32x2{v, v}; }
#define FS_ROW(cells, i) ((cells)[(i) / 2][(i) % 2])
#endif

struct FusedSmallPlan {
  uint32_t column_of_slot[FS_COLUMNS];   // index into FusedPlan::columns, 0xFFFFFFFF: the slot is empty
  uint32_t slot_of_column[FS_COLUMNS];   // ... and back (FusedNode::column -> slot)
  uint32_t filter_width[FS_FILTERS];     // bytes per value id
  uint32_t n_inputs;                     // accumulators 0 .. n_inputs - 1 have an expression; the others are COUNT(*)
  uint32_t n_nodes;                      // four bits per input: the nodes of program[d] -- without its first nodes where those are the whole of input d - 1:
                                         // the inputs are evaluated in order on one stack, input d then starts on what input d - 1 left
  uint32_t literal[FS_LITERALS];         // float bits
  uint64_t program[FS_INPUTS];           // the input's postfix nodes, five bits each (FS_*), first node lowest -- kernel arguments: scalar registers
};
__host__ __device__ constexpr size_t fs_lds_bytes() { return size_t{FS_WINDOW} * 4 + size_t{FS_NARROW} * 256 * 4; }

// four consecutive value ids as loaded (1-byte ids: x; 2-byte ids: x, y)
#ifndef HY_FS_CACHED_IDS   // value ids are read once: nontemporal loads leave the L2 to the dictionary lines of the gathers (0.569 -> 0.552 ms; the switch is for timing)
__device__ __forceinline__ uint32_t fs_load_ids8(const void* data, uint32_t row) { return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(static_cast<const char*>(data) + row)); }
__device__ __forceinline__ u32x2 fs_load_ids16(const void* data, uint32_t row) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(static_cast<const char*>(data) + size_t{row} * 2)); }
#else
__device__ __forceinline__ uint32_t fs_load_ids8(const void* data, uint32_t row) { return *(const __attribute__((address_space(1))) uint32_t*)(static_cast<const char*>(data) + row); }
__device__ __forceinline__ u32x2 fs_load_ids16(const void* data, uint32_t row) { return *(const global_u32x2*)(static_cast<const char*>(data) + size_t{row} * 2); }
#endif
__device__ __forceinline__ uint32_t fs_id8(uint32_t v, int j) { return (v >> (8 * j)) & 0xFFu; }
__device__ __forceinline__ uint32_t fs_id16(const u32x2& v, int j) { return ((j < 2 ? v.x : v.y) >> (16 * (j & 1))) & 0xFFFFu; }

__global__ __launch_bounds__(FS_THREADS) void fused_small_domain(AggArgs a, const FusedPlan* __restrict__ plan, FusedSmallPlan lean, uint32_t n_chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fs_smem[];
  float* s_window = reinterpret_cast<float*>(fs_smem);                                          // [FS_WINDOW]
  float (*s_dict)[256] = reinterpret_cast<float (*)[256]>(fs_smem + size_t{FS_WINDOW} * 4);     // [FS_NARROW][256]: dictionaries of the 1-byte columns
  __shared__ ScanJob s_jobs[FS_FILTERS];
  __shared__ uint32_t s_dense_of_code[SD_CODES];               // (as in aggregate_small_domain)
  __shared__ uint32_t s_code_of_dense[SD_CODES];
  __shared__ uint32_t s_n_dense, s_refused;
  __shared__ __attribute__((aligned(8))) uint32_t s_dense_map[4];
  __shared__ double s_sum[SD_DENSE][FS_INPUTS];
  __shared__ uint32_t s_rows[SD_DENSE], s_first[SD_DENSE], s_last[SD_DENSE];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const uint64_t chunk_base = a.row_base[chunk];
  const uint32_t rows = static_cast<uint32_t>(a.row_base[chunk + 1] - chunk_base);
  const uint32_t n_filters = plan->n_filters, n_inputs = lean.n_inputs;

  // ---- descriptors -----------------------------------------------------------------------------------------------------------
  const void* key_data[SD_KEYS];
  uint32_t key_stride[SD_KEYS], key_size[SD_KEYS];
  uint32_t product = 1;
#pragma unroll
  for (uint32_t g = 0; g < SD_KEYS; ++g) {
    key_data[g] = nullptr;
    key_stride[g] = key_size[g] = 0;
    if (g < a.n_groupby) {
      const DevSegment seg = a.groupby[g].segments[chunk];
      key_data[g] = seg.data;
      key_size[g] = seg.aux_size;
      key_stride[g] = product;
      product *= seg.aux_size + 1;   // + 1: the NULL value id
    }
  }
  const void* filter_data[FS_FILTERS];
#pragma unroll
  for (uint32_t f = 0; f < FS_FILTERS; ++f) {
    filter_data[f] = nullptr;
    if (f < n_filters) filter_data[f] = plan->filters[f].segments[chunk].data;
  }
  const void* narrow_data[FS_NARROW];
  uint32_t narrow_size[FS_NARROW];
#pragma unroll
  for (uint32_t c = 0; c < FS_NARROW; ++c) {
    narrow_data[c] = nullptr;
    narrow_size[c] = 0;
    if (lean.column_of_slot[c] != 0xFFFFFFFFu) {
      const DevSegment seg = plan->columns[lean.column_of_slot[c]][chunk];
      narrow_data[c] = seg.data;
      narrow_size[c] = seg.aux_size;
      if (tid < 256) s_dict[c][tid] = tid < seg.aux_size ? ((const global_f32*)seg.aux)[tid] : 0.0f;
    }
  }
  const bool has_wide = lean.column_of_slot[FS_NARROW] != 0xFFFFFFFFu;
  const void* wide_data = nullptr;
  const void* wide_dictionary = nullptr;
  uint32_t wide_size = 0;
  if (has_wide) {
    const DevSegment seg = plan->columns[lean.column_of_slot[FS_NARROW]][chunk];
    wide_data = seg.data;
    wide_dictionary = seg.aux;
    wide_size = seg.aux_size;
  }
  if (tid < FS_FILTERS && tid < n_filters) s_jobs[tid] = plan->filters[tid].jobs[chunk];
  if (tid < SD_CODES) { s_dense_of_code[tid] = 0xFFu; s_code_of_dense[tid] = 0; }
  if (tid < SD_DENSE) {
    s_rows[tid] = 0; s_first[tid] = 0xFFFFFFFFu; s_last[tid] = 0;
    for (uint32_t d = 0; d < FS_INPUTS; ++d) s_sum[tid][d] = 0.0;
  }
  if (tid == 0) { s_n_dense = 0; s_refused = 0; }
  if (tid < 4) s_dense_map[tid] = 0;
  __syncthreads();

  // the filters' jobs of this chunk: value-id ranges (lo, span, the NULL id, inverted or not); a job that takes every row is skipped
  uint32_t job_lo[FS_FILTERS], job_span[FS_FILTERS], job_null[FS_FILTERS];
  bool job_invert[FS_FILTERS], job_reads[FS_FILTERS];
  bool nothing = false, refused = false;
#pragma unroll
  for (uint32_t f = 0; f < FS_FILTERS; ++f) {
    job_lo[f] = job_span[f] = 0; job_null[f] = 0xFFFFFFFFu; job_invert[f] = false; job_reads[f] = false;
    if (f >= n_filters) continue;
    const ScanJob job = uniform(s_jobs[f]);
    if (job.mode == JOB_ALL) continue;
    if (job.mode == JOB_NONE || (job.flags & JF_NEVER)) { nothing = true; continue; }
    if (job.mode != JOB_SCAN || job.kind == KIND_VALUE_ID_SET || job.kind == KIND_VISIBLE) { refused = true; continue; }
    job_lo[f] = static_cast<uint32_t>(job.lo); job_span[f] = static_cast<uint32_t>(job.span); job_null[f] = job.null_vid;
    job_invert[f] = job.flags & JF_INVERT;
    job_reads[f] = true;
  }
  if (refused || rows > FS_SPAN) {
    if (tid == 0) __hip_atomic_store(&a.overflow[FLAG_SMALL_REFUSED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (nothing || rows == 0) return;

  double acc[FS_INPUTS][SD_DENSE];
#pragma unroll
  for (uint32_t d = 0; d < FS_INPUTS; ++d) {
#pragma unroll
    for (uint32_t k = 0; k < SD_DENSE; ++k) acc[d][k] = 0.0;
  }
  uint32_t rows_of[SD_DENSE], ends_of[SD_DENSE];   // ends_of: the group's first row | its last row << 16 (chunk offsets below 65536)
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) rows_of[k] = ends_of[k] = 0;
  uint32_t bad = 0;   // a NULL in an input column of a row that passed the filters

  // The 2-byte column's dictionary (l_extendedprice: 240 KB per chunk): its first FS_WINDOW entries are staged in LDS, the others are
  // read where they are -- the chunk's dictionary is touched by this workgroup only: its lines stay in this XCD's L2.
  const uint32_t in_window = has_wide ? min(FS_WINDOW, wide_size) : 0u;
  {   // (16-byte loads, all of a thread's requested before the first is stored: one round trip to HBM)
    constexpr uint32_t BATCH = (FS_WINDOW / 4 + FS_THREADS - 1) / FS_THREADS;
    u32x4 staged[BATCH];
#pragma unroll
    for (uint32_t n = 0; n < BATCH; ++n) {
      const uint32_t i = (n * FS_THREADS + tid) * 4;
      staged[n] = u32x4{0, 0, 0, 0};
      if (i < in_window) staged[n] = *(const global_u32x4*)(static_cast<const char*>(wide_dictionary) + size_t{i} * 4);   // (ends inside the buffer's padding)
    }
#pragma unroll
    for (uint32_t n = 0; n < BATCH; ++n) {
      const uint32_t i = (n * FS_THREADS + tid) * 4;
      if (i < in_window) *reinterpret_cast<u32x4*>(&s_window[i]) = staged[n];
    }
  }
  __syncthreads();
  {
#pragma unroll 1
    for (uint32_t step = 0; step < FS_STEPS; ++step) {
      const uint32_t first = (step * FS_THREADS + tid) * FS_ROWS;
      if (step * FS_THREADS * FS_ROWS >= rows) break;
      const uint32_t at = first < rows ? first : 0u;   // (a lane without rows reads the chunk's first ids)
      // ---- loads: four ids of every column -------------------------------------------------------------------------------------
      uint32_t key_ids[SD_KEYS], narrow_ids[FS_NARROW];
      u32x2 filter_ids[FS_FILTERS], wide_ids = u32x2{0, 0};
#pragma unroll
      for (uint32_t g = 0; g < SD_KEYS; ++g) key_ids[g] = g < a.n_groupby ? fs_load_ids8(key_data[g], at) : 0u;
#pragma unroll
      for (uint32_t f = 0; f < FS_FILTERS; ++f) {
        filter_ids[f] = u32x2{0, 0};
        if (job_reads[f]) {
          if (lean.filter_width[f] == 1) filter_ids[f].x = fs_load_ids8(filter_data[f], at);
          else filter_ids[f] = fs_load_ids16(filter_data[f], at);
        }
      }
#pragma unroll
      for (uint32_t c = 0; c < FS_NARROW; ++c) narrow_ids[c] = narrow_data[c] ? fs_load_ids8(narrow_data[c], at) : 0u;
      if (has_wide) wide_ids = fs_load_ids16(wide_data, at);
      // ---- the filters ----------------------------------------------------------------------------------------------------------
      uint32_t pass = 0;
#pragma unroll
      for (int j = 0; j < static_cast<int>(FS_ROWS); ++j) pass |= (first + j < rows ? 1u : 0u) << j;
#pragma unroll
      for (uint32_t f = 0; f < FS_FILTERS; ++f) {
        if (!job_reads[f]) continue;
        uint32_t matches = 0;
#pragma unroll
        for (int j = 0; j < static_cast<int>(FS_ROWS); ++j) {
          const uint32_t id = lean.filter_width[f] == 1 ? fs_id8(filter_ids[f].x, j) : fs_id16(filter_ids[f], j);
          matches |= ((((id - job_lo[f]) <= job_span[f]) != job_invert[f]) && id != job_null[f] ? 1u : 0u) << j;
        }
        pass &= matches;
      }
      // ---- groups: codes -> dense indices (a nibble per row; 0xF: the row does not count) -----------------------------------------
      uint32_t codes = 0;
#pragma unroll
      for (int j = 0; j < static_cast<int>(FS_ROWS); ++j) {
        uint32_t code = 0;
#pragma unroll
        for (uint32_t g = 0; g < SD_KEYS; ++g) {   // (a GROUP BY column that is not there: ids 0, stride 0)
          const uint32_t id = fs_id8(key_ids[g], j);
          code += (id < key_size[g] ? id : key_size[g]) * key_stride[g];
        }
        codes |= (code & 0xFu) << (4 * j);
      }
      uint32_t group[FS_ROWS];    // dense index of the row, 0xF: not counted
      {
        uint32_t assigned = __hip_atomic_load(&s_dense_map[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (not a volatile read: that waits for every outstanding load, aggregate_small.hpp)
        uint32_t wanted = 0;   // codes of this lane's rows that passed
#pragma unroll
        for (int j = 0; j < static_cast<int>(FS_ROWS); ++j) {
          if ((pass >> j) & 1) wanted |= 1u << ((codes >> (4 * j)) & 0xFu);
        }
        while (__any((wanted & ~assigned) != 0)) {   // (the chunk's first rows only: see aggregate_small_domain)
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
          assigned = __hip_atomic_load(&s_dense_map[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const uint64_t map = static_cast<uint64_t>(__hip_atomic_load(&s_dense_map[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) |
                             static_cast<uint64_t>(__hip_atomic_load(&s_dense_map[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) << 32;
#pragma unroll
        for (int j = 0; j < static_cast<int>(FS_ROWS); ++j) {
          const uint32_t code = (codes >> (4 * j)) & 0xFu;
          const uint32_t d = static_cast<uint32_t>(map >> (4 * code)) & 0xFu;
          group[j] = ((pass >> j) & 1) && d < SD_DENSE ? d : 0xFu;
        }
      }
      // ---- rows, first and last row per dense group ------------------------------------------------------------------------------------
      {
#pragma unroll
        for (uint32_t k = 0; k < SD_DENSE; ++k) {
          uint32_t hits = 0;
#pragma unroll
          for (int j = 0; j < static_cast<int>(FS_ROWS); ++j) hits |= (group[j] == k ? 1u : 0u) << j;
          if (hits) {
            if (rows_of[k] == 0) ends_of[k] = first + (__ffs(hits) - 1);
            ends_of[k] = (ends_of[k] & 0xFFFFu) | (first + (31 - __clz(hits))) << 16;
            rows_of[k] += __popc(hits);
          }
        }
      }
      // ---- the columns' values: dictionaries in LDS, the 2-byte column's entries behind the window through the L2 ---------------------
      fs_cell value[FS_COLUMNS][FS_CELLS];
      if (has_wide) {
        uint32_t id[FS_ROWS];
        float far[FS_ROWS];
#pragma unroll
        for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) {
          id[i] = fs_id16(wide_ids, i);
          far[i] = 0.0f;
          if (id[i] >= in_window && id[i] < wide_size) far[i] = ((const global_f32*)wide_dictionary)[id[i]];
        }
        if (max(max(id[0], id[1]), max(id[2], id[3])) >= wide_size) {   // a NULL id? (rare: the four ids are looked at one by one only then)
#pragma unroll
          for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) bad |= ((pass >> i) & 1) && id[i] >= wide_size ? 1u : 0u;   // (rows behind the chunk's end read padding)
        }
#pragma unroll
        for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) {
          const float near = s_window[id[i] < in_window ? id[i] : 0u];
          FS_ROW(value[FS_NARROW], i) = id[i] < in_window ? near : far[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) FS_ROW(value[FS_NARROW], i) = 0.0f;
      }
#pragma unroll
      for (uint32_t c = 0; c < FS_NARROW; ++c) {
        uint32_t largest = 0;
#pragma unroll
        for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) {
          FS_ROW(value[c], i) = 0.0f;
          if (!narrow_data[c]) continue;
          const uint32_t id = fs_id8(narrow_ids[c], i);
          largest = max(largest, id);
          FS_ROW(value[c], i) = s_dict[c][id];
        }
        if (narrow_data[c] && largest >= narrow_size[c]) {
#pragma unroll
          for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) bad |= ((pass >> i) & 1) && fs_id8(narrow_ids[c], i) >= narrow_size[c] ? 1u : 0u;
        }
      }
      // ---- the expressions -----------------------------------------------------------------------------------------------------------
      // a three-slot stack in registers (slot 0 = top), as evaluate_input's -- floats only, two rows to a cell (v_pk_add_f32 / v_pk_mul_f32)
      fs_cell s0[FS_CELLS], s1[FS_CELLS], s2[FS_CELLS];
#pragma unroll
      for (int h = 0; h < FS_CELLS; ++h) s0[h] = s1[h] = s2[h] = fs_cell_of(0.0f);
#pragma unroll
      for (uint32_t d = 0; d < FS_INPUTS; ++d) {
        if (d >= n_inputs) continue;
        const uint32_t n_nodes = (lean.n_nodes >> (4 * d)) & 0xFu;
        uint64_t program = lean.program[d];
#pragma unroll 1
        for (uint32_t n = 0; n < n_nodes; ++n, program >>= 5) {
          const uint32_t node = static_cast<uint32_t>(program) & 31u;
          if (node >= FS_ADD) {
            if (node == FS_ADD) {
#pragma unroll
              for (int h = 0; h < FS_CELLS; ++h) s0[h] = s1[h] + s0[h];
            } else if (node == FS_SUB) {
#pragma unroll
              for (int h = 0; h < FS_CELLS; ++h) s0[h] = s1[h] - s0[h];
            } else {
#pragma unroll
              for (int h = 0; h < FS_CELLS; ++h) s0[h] = s1[h] * s0[h];
            }
#pragma unroll
            for (int h = 0; h < FS_CELLS; ++h) s1[h] = s2[h];
          } else {
#pragma unroll
            for (int h = 0; h < FS_CELLS; ++h) { s2[h] = s1[h]; s1[h] = s0[h]; }
            if (node < FS_PUSH_LITERAL) {
#pragma unroll
              for (uint32_t c = 0; c < FS_COLUMNS; ++c) {
                if (node != c) continue;
#pragma unroll
                for (int h = 0; h < FS_CELLS; ++h) s0[h] = value[c][h];
              }
            } else {
              const uint32_t which = node - FS_PUSH_LITERAL;
              const float literal = __uint_as_float(which == 0 ? lean.literal[0] : which == 1 ? lean.literal[1] : which == 2 ? lean.literal[2] : lean.literal[3]);
#pragma unroll
              for (int h = 0; h < FS_CELLS; ++h) s0[h] = fs_cell_of(literal);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < static_cast<int>(FS_ROWS); ++i) {
          const double x = static_cast<double>(FS_ROW(s0, i));
#pragma unroll
          for (uint32_t k = 0; k < SD_DENSE; ++k) acc[d][k] += group[i] == k ? x : 0.0;
        }
      }
    }
  }
  if (__any(bad != 0) && lane == 0) s_refused = 1;

  // ---- the chunk's groups -------------------------------------------------------------------------------------------------------
#pragma unroll
  for (uint32_t k = 0; k < SD_DENSE; ++k) {
    const uint32_t n = wave_reduce_u32_to_lane63(rows_of[k], 0u, false, false);
    const uint32_t first = wave_reduce_u32_to_lane63(rows_of[k] ? ends_of[k] & 0xFFFFu : 0xFFFFFFFFu, 0xFFFFFFFFu, true, false);
    const uint32_t last = wave_reduce_u32_to_lane63(ends_of[k] >> 16, 0u, false, true);
    if (lane == 63 && n) {
      atomicAdd(&s_rows[k], n);
      atomicMin(&s_first[k], first);
      atomicMax(&s_last[k], last);
    }
#pragma unroll
    for (uint32_t d = 0; d < FS_INPUTS; ++d) {
      if (d >= n_inputs) continue;
      const uint64_t sum = wave_reduce_to_lane63(static_cast<uint64_t>(__double_as_longlong(acc[d][k])), 0ull, [](uint64_t x, uint64_t y) {
        return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) + __longlong_as_double(static_cast<long long>(y))));
      });
      if (lane == 63 && n) atomicAdd(&s_sum[k][d], __longlong_as_double(static_cast<long long>(sum)));
    }
  }
  __syncthreads();
  const uint32_t n_dense = s_n_dense;
  if (s_refused || n_dense > SD_DENSE) {   // a NULL input, a fifth group: fused_rows takes the query
    if (tid == 0) __hip_atomic_store(&a.overflow[FLAG_SMALL_REFUSED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (tid == 0) {
    uint32_t passed = 0;
    for (uint32_t k = 0; k < SD_DENSE; ++k) passed += s_rows[k];
    if (passed) atomicAdd(reinterpret_cast<unsigned long long*>(a.overflow + FLAG_PASSED), static_cast<unsigned long long>(passed));
  }
  // merge into the global table: thread = dense index
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
    if (g >= n_inputs) { merge_global(a, gslot, g, 0, s_rows[tid]); continue; }
    uint64_t bits = 0;
#pragma unroll
    for (uint32_t d = 0; d < FS_INPUTS; ++d) if (d == g) bits = static_cast<uint64_t>(__double_as_longlong(s_sum[tid][d]));
    merge_global(a, gslot, g, bits, s_rows[tid]);
  }
}
