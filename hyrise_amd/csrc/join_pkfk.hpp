// join_pkfk.hpp -- the primary-key / foreign-key probe of JoinHash (included by join.hip, inside namespace hy).
//
// What it replaces (reference, CPU): probe / probe_semi_anti over the radix-partitioned probe side,
// operators/join_hash/join_hash_steps.hpp:624-922, with the output order of :541-591, :655-760 (partition, probe row,
// build row; a new PosList every 131 070 materialised probe elements).
//
// When it runs: the build side is a rank table (unique integer keys, struct RankTable in join.hip) and every segment of
// the probe column is one a SliceView describes (int32 values / FrameOfReference offsets, no NULLs) -- TPC-H
// orders x lineitem (config 3), the first join of an SSB star plan.  Everything else keeps the general kernels.
//
// Four launches, no host round trip in between:
//   pk_count   one workgroup per 8192-row tile (= slice): the tile's rows with 16-byte loads, one 8-byte rank-table
//              entry per row, materialised elements | pairs << 16 per (partition, tile) in LDS cells
//   pk_scan    one workgroup per partition: exclusive prefix of its row of counts (elements and pairs); the workgroup
//              that finishes last plans the output (partition origins, PosLists per group, capacity check, mailbox)
//   pk_emit    one workgroup per tile: re-evaluates the rows (one dependent load per row: cheaper than handing 6 bytes
//              per row from pass 1 to pass 2), ranks the pairs of a partition with ONE returning LDS atomic per pair,
//              stages them partition by partition in LDS and writes contiguous runs with 16-byte stores.  8192-row
//              tiles make the runs 2 KB at config 3 (32 of the 128 partitions are populated): the tile size is the
//              lever on the write rate, tools/hbm_write.hip
//   pk_cuts    one workgroup per output PosList: the row of its first element
// HBM traffic at config 3: 2 x probe keys + 16 B per pair + 3 x 4 B per (partition, tile) cell.
#pragma once

constexpr uint32_t PK_TILE = SLICE_ROWS;                 // 8192 rows: a probe tile is a slice
constexpr uint32_t PK_THREADS = 512;
constexpr uint32_t PK_WAVES = PK_THREADS / 64;           // 8
constexpr uint32_t PK_WAVE_ROWS = PK_TILE / PK_WAVES;    // 1024 consecutive rows per wave
constexpr uint32_t PK_ROUNDS = PK_WAVE_ROWS / 64;        // 16 rows per lane
constexpr uint32_t PK_COUNT_THREADS = 256;
constexpr uint32_t PK_COUNT_WAVE_ROWS = PK_TILE / (PK_COUNT_THREADS / 64);   // 2048: one FrameOfReference block per wave
constexpr uint32_t PK_COUNT_BATCHES = PK_COUNT_WAVE_ROWS / 512;              // 4 batches of 512 rows, eight consecutive rows per lane
constexpr uint32_t PK_SCAN_THREADS = 256;
constexpr uint32_t PK_SCAN_CHUNK = 4096;                 // tiles per pass of pk_scan's loop
static_assert(HY_FOR_BLOCK_SIZE % PK_WAVE_ROWS == 0 && HY_FOR_BLOCK_SIZE == PK_COUNT_WAVE_ROWS, "a wave's rows must lie in one FrameOfReference block");
static_assert(PK_TILE <= (1u << 13), "a staged pair keeps its row in 13 bits");

struct PkArgs {
  const SliceView* views;          // [n_tiles] the probe column's slices
  uint32_t n_tiles;
  uint32_t stride;                 // row stride of the [P][stride] arrays below (> n_tiles: entry n_tiles of a row is its total)
  uint32_t mode;                   // HY_JOIN_*
  uint32_t radix_bits;
  uint32_t keep_nulls;             // probe side keeps NULLs (outer / anti modes): the build side's Bloom filter is not applied
  uint32_t n_groups;               // output groups: the partitions, or (radix_bits == 0) the probe chunks
  const uint8_t* build_bloom;      // decides which partner-less probe rows count as materialised, or nullptr
  RankTable rank;
  const uint32_t* ids32;           // RowIDs by rank: packed, or ...
  const hy_row_id* row_ids;        // ... plain (neither: rank.identity_rows)
  uint32_t* counts;                // [P][stride] pk_count: elements | pairs << 16
  uint32_t* rel_elements;          // [P][stride] pk_scan: elements / pairs of the partition in earlier tiles
  uint32_t* rel_pairs;
  uint32_t* totals;                // [2][P] elements, pairs of a partition
  uint64_t* origin_pairs;          // [P + 1] pairs of earlier partitions
  uint32_t* slice_base;            // [n_groups + 1] first output PosList of a group
  const uint32_t* group_first_tile;   // radix_bits == 0: [n_chunks + 1] first tile of every probe chunk
  uint32_t* ticket;                // [0] pk_scan's arrival counter
  JoinPlan* plan;
  JoinMailbox* mailbox;
  uint64_t capacity;
  uint32_t slice_capacity;
  uint32_t plain_stores;           // debug (HY_JOIN_PLAIN_STORES): write-back stores instead of nontemporal ones in pk_emit
  hy_row_id* build_out;            // nullptr: Semi / Anti
  hy_row_id* probe_out;
  uint64_t* slice_offsets;
};

// XCD x = blockIdx % 8 takes the x-th eighth of the tiles (block_tile in join.hip): neighbouring tiles, whose output runs are
// neighbours in memory, meet in one L2.
__device__ __forceinline__ uint32_t pk_block_tile(uint32_t n_tiles) { return (blockIdx.x & 7) * ((n_tiles + 7) / 8) + (blockIdx.x >> 3); }

// Output pairs of a probe row without NULL key (pairs_of in join.hip with is_null == false).
template <bool INNER>
__device__ __forceinline__ bool pk_emits(uint32_t mode, bool found, bool* null_partner) {
  if constexpr (INNER) { *null_partner = false; return found; }
  const bool outer = mode == HY_JOIN_LEFT || mode == HY_JOIN_RIGHT;
  *null_partner = outer && !found;
  return outer ? true : (mode == HY_JOIN_INNER || mode == HY_JOIN_SEMI) ? found : !found;
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------------------
template <uint32_t WIDTH>
__device__ __forceinline__ void pk_count_wave(const PkArgs& a, const SliceView& view, uint32_t wave, uint32_t lane, uint32_t* cells) {
  const char* base = static_cast<const char*>(view.data);
  const uint32_t row_count = view.row_count;
  const uint32_t wave_first = wave * PK_COUNT_WAVE_ROWS;
  if (wave_first >= row_count) return;
  u32x4_t words[PK_COUNT_BATCHES][2];
#pragma unroll
  for (uint32_t b = 0; b < PK_COUNT_BATCHES; ++b) {
    const uint32_t first = wave_first + b * 512 + lane * 8;
    load_batch_words<WIDTH>(base, view.row_begin + (first < row_count ? first : 0), words[b]);
  }
  const uint32_t bias = view.kind == VIEW_INT32 ? 0u : static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[(view.row_begin + wave_first) / HY_FOR_BLOCK_SIZE]);
  const uint32_t origin = static_cast<uint32_t>(a.rank.key_min), range = static_cast<uint32_t>(a.rank.range);
  const uint32_t mask = a.radix_bits ? (1u << a.radix_bits) - 1 : 0u;
#pragma unroll
  for (uint32_t b = 0; b < PK_COUNT_BATCHES; ++b) {
    uint32_t low[8];
    u32x2_t entry[8];
    uint32_t valid = 0, look = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      const bool in = wave_first + b * 512 + lane * 8 + j < row_count;
      low[j] = batch_word<WIDTH>(words[b], j) + bias;
      const uint32_t distance = low[j] - origin;   // (32-bit: both sides' keys are int32 values, see pk_path_applies)
      const bool looked = in && distance <= range;
      entry[j] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(a.rank.entries) + (looked ? (distance >> 5) * 8u : 0u));
      valid |= (in ? 1u : 0u) << j;
      look |= (looked ? 1u : 0u) << j;
    }
    uint32_t found = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      if (((look >> j) & 1) && ((entry[j].x >> ((low[j] - origin) & 31)) & 1)) found |= 1u << j;
    }
    if (a.build_bloom && !a.keep_nulls && __any((valid & ~found) != 0)) {   // partner-less rows: materialised only if the build side's filter has their bit
      uint32_t miss = 0;
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        if (((valid & ~found) >> j) & 1) miss |= (a.build_bloom[low[j] & (BLOOM_BITS - 1)] == 0 ? 1u : 0u) << j;
      }
      valid &= ~miss;
    }
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      if (!((valid >> j) & 1)) continue;
      bool null_partner;
      const bool emit = pk_emits<false>(a.mode, (found >> j) & 1, &null_partner);
      atomicAdd(&cells[(low[j] & mask) * COUNT_COPIES + ((lane + j) & (COUNT_COPIES - 1))], emit ? 0x10001u : 1u);
    }
  }
}

__global__ __launch_bounds__(PK_COUNT_THREADS) void pk_count(PkArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cells[MAX_PARTITIONS * COUNT_COPIES];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t tile = pk_block_tile(a.n_tiles);
  if (blockIdx.x == 0 && tid == 0) *a.ticket = 0;   // pk_scan's arrival counter (pk_scan runs behind this kernel)
  if (tile >= a.n_tiles) return;
  for (uint32_t i = tid; i < partitions * COUNT_COPIES; i += PK_COUNT_THREADS) s_cells[i] = 0;
  __syncthreads();
  const SliceView view = a.views[tile];
  if (view.kind == VIEW_FOR8) pk_count_wave<1>(a, view, wave, lane, s_cells);
  else if (view.kind == VIEW_FOR16) pk_count_wave<2>(a, view, wave, lane, s_cells);
  else pk_count_wave<4>(a, view, wave, lane, s_cells);
  __syncthreads();
  for (uint32_t partition = tid; partition < partitions; partition += PK_COUNT_THREADS) {
    const u32x4_t low = *reinterpret_cast<const u32x4_t*>(s_cells + partition * COUNT_COPIES), high = *reinterpret_cast<const u32x4_t*>(s_cells + partition * COUNT_COPIES + 4);
    a.counts[static_cast<size_t>(partition) * a.stride + tile] = low.x + low.y + low.z + low.w + high.x + high.y + high.z + high.w;
  }
}

// ---- scan + plan ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t pk_wave_inclusive_scan64(uint64_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t other = __shfl_up(v, d, 64);
    if (lane >= static_cast<uint32_t>(d)) v += other;
  }
  return v;
}

// Exclusive prefix of `v` over the PK_SCAN_THREADS threads of the workgroup (s_tmp: 5 words); *total = the sum.  Ends with a barrier.
__device__ __forceinline__ uint64_t pk_block_exclusive_scan64(uint64_t v, uint64_t* s_tmp, uint32_t tid, uint64_t* total) {
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint64_t inclusive = pk_wave_inclusive_scan64(v, lane);
  __syncthreads();   // (s_tmp may still be read from an earlier call)
  if (lane == 63) s_tmp[wave] = inclusive;
  __syncthreads();
  uint64_t before = 0, all = 0;
  for (uint32_t w = 0; w < PK_SCAN_THREADS / 64; ++w) { if (w < wave) before += s_tmp[w]; all += s_tmp[w]; }
  *total = all;
  return before + inclusive - v;
}

// The plan of the output (plan_output in join.hip), by pk_scan's last workgroup.
__device__ void pk_plan(const PkArgs& a, uint64_t* s_tmp, uint32_t tid) {
  const uint32_t partitions = 1u << a.radix_bits;
  uint64_t n_pairs = 0;
  uint32_t n_slices = 0;
  if (a.radix_bits) {   // groups = partitions (<= 256 = PK_SCAN_THREADS)
    const uint64_t elements = tid < partitions ? a.totals[tid] : 0, pairs = tid < partitions ? a.totals[partitions + tid] : 0;
    uint64_t total = 0;
    const uint64_t pairs_before = pk_block_exclusive_scan64(pairs, s_tmp, tid, &total);
    n_pairs = total;
    const uint64_t slices = (elements + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK;
    const uint64_t slices_before = pk_block_exclusive_scan64(slices, s_tmp, tid, &total);
    n_slices = static_cast<uint32_t>(total);
    if (tid < partitions) {
      a.origin_pairs[tid] = pairs_before;
      a.slice_base[tid] = static_cast<uint32_t>(slices_before);
    }
    if (tid == 0) { a.origin_pairs[partitions] = n_pairs; a.slice_base[partitions] = n_slices; }
  } else {   // groups = probe chunks: their tiles are consecutive in the one row of counts
    n_pairs = a.totals[1];
    uint64_t running = 0;
    for (uint32_t begin = 0; begin < a.n_groups; begin += PK_SCAN_THREADS) {
      const uint32_t g = begin + tid;
      uint64_t slices = 0;
      if (g < a.n_groups) {
        const uint32_t elements = a.rel_elements[a.group_first_tile[g + 1]] - a.rel_elements[a.group_first_tile[g]];
        slices = (elements + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK;
      }
      uint64_t total = 0;
      const uint64_t before = pk_block_exclusive_scan64(slices, s_tmp, tid, &total);
      if (g < a.n_groups) a.slice_base[g] = static_cast<uint32_t>(running + before);
      running += total;
    }
    n_slices = static_cast<uint32_t>(running);
    if (tid == 0) { a.origin_pairs[0] = 0; a.origin_pairs[1] = n_pairs; a.slice_base[a.n_groups] = n_slices; }
  }
  if (tid == 0) {
    const uint32_t fits = n_pairs <= a.capacity && n_slices <= a.slice_capacity ? 1u : 0u;
    a.plan->fits = fits;
    a.plan->n_slices = n_slices;
    if (fits && a.slice_offsets) a.slice_offsets[n_slices] = n_pairs;
    a.mailbox->n_pairs = n_pairs;
    a.mailbox->n_slices = n_slices;
    a.mailbox->n_uncached = 0;
    a.mailbox->fits = fits;
    __threadfence_system();
  }
}

__global__ __launch_bounds__(PK_SCAN_THREADS) void pk_scan(PkArgs a) {
  // element i of a chunk sits at i + i / 16: a thread's 16 consecutive elements start 17 words apart -> no bank conflicts
  constexpr uint32_t PER_THREAD = PK_SCAN_CHUNK / PK_SCAN_THREADS, PADDED = PK_SCAN_CHUNK + PK_SCAN_CHUNK / PER_THREAD;
  __shared__ uint32_t s_elements[PADDED], s_pairs[PADDED];
  __shared__ uint64_t s_tmp[8];
  __shared__ uint32_t s_last;
  const uint32_t tid = threadIdx.x, partition = blockIdx.x;
  const uint32_t* counts = a.counts + static_cast<size_t>(partition) * a.stride;
  uint32_t* rel_elements = a.rel_elements + static_cast<size_t>(partition) * a.stride;
  uint32_t* rel_pairs = a.rel_pairs + static_cast<size_t>(partition) * a.stride;
  uint64_t carry = 0;   // elements | pairs << 32 of the chunks before
  for (uint32_t begin = 0; begin < a.n_tiles; begin += PK_SCAN_CHUNK) {
    const uint32_t m = a.n_tiles - begin < PK_SCAN_CHUNK ? a.n_tiles - begin : PK_SCAN_CHUNK;
    for (uint32_t i = tid; i < PK_SCAN_CHUNK; i += PK_SCAN_THREADS) {
      const uint32_t packed = i < m ? counts[begin + i] : 0u;
      s_elements[i + i / PER_THREAD] = packed & 0xFFFFu;
      s_pairs[i + i / PER_THREAD] = packed >> 16;
    }
    __syncthreads();
    uint32_t* mine_elements = s_elements + tid * (PER_THREAD + 1);
    uint32_t* mine_pairs = s_pairs + tid * (PER_THREAD + 1);
    uint64_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER_THREAD; ++j) sum += static_cast<uint64_t>(mine_elements[j]) | static_cast<uint64_t>(mine_pairs[j]) << 32;
    uint64_t total = 0;
    uint64_t run = carry + pk_block_exclusive_scan64(sum, s_tmp, tid, &total);   // (neither half overflows: both count probe rows, < 2^32)
#pragma unroll
    for (uint32_t j = 0; j < PER_THREAD; ++j) {
      const uint64_t v = static_cast<uint64_t>(mine_elements[j]) | static_cast<uint64_t>(mine_pairs[j]) << 32;
      mine_elements[j] = static_cast<uint32_t>(run);
      mine_pairs[j] = static_cast<uint32_t>(run >> 32);
      run += v;
    }
    __syncthreads();
    for (uint32_t i = tid; i < m; i += PK_SCAN_THREADS) {
      rel_elements[begin + i] = s_elements[i + i / PER_THREAD];
      rel_pairs[begin + i] = s_pairs[i + i / PER_THREAD];
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0) {
    rel_elements[a.n_tiles] = static_cast<uint32_t>(carry);
    rel_pairs[a.n_tiles] = static_cast<uint32_t>(carry >> 32);
    a.totals[partition] = static_cast<uint32_t>(carry);
    a.totals[gridDim.x + partition] = static_cast<uint32_t>(carry >> 32);
  }
  // the workgroup that arrives last sees every partition's totals: it plans the output
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t arrived = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = arrived + 1 == gridDim.x ? 1u : 0u;
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  pk_plan(a, s_tmp, tid);
}

// ---- evaluation of a tile's rows (pass 2, cuts) ----------------------------------------------------------------------------------
// Wave w owns rows [1024 w, 1024 (w + 1)) of the tile, row k * 64 + lane in round k: row order = (wave, round, lane).
// meta[k] = partition | null_partner << 9 | emit << 10 (INVALID_PARTITION: the row is not materialised); rank[k] = the partner's rank.
template <bool INNER>
__device__ __forceinline__ void pk_evaluate(const PkArgs& a, const SliceView& view, uint32_t wave, uint32_t lane, uint32_t (&meta)[PK_ROUNDS], uint32_t (&rank)[PK_ROUNDS]) {
  const uint32_t row_count = view.row_count;
  const uint32_t wave_first = wave * PK_WAVE_ROWS;
  uint32_t raw[PK_ROUNDS];
  if (row_count == PK_TILE) {   // a full tile: one address per lane, immediate offsets
    const uint32_t first = view.row_begin + wave_first + lane;
    if (view.kind == VIEW_FOR16) {
      const uint16_t* base = static_cast<const uint16_t*>(view.data) + first;
#pragma unroll
      for (uint32_t k = 0; k < PK_ROUNDS; ++k) raw[k] = base[k * 64];
    } else if (view.kind == VIEW_FOR8) {
      const uint8_t* base = static_cast<const uint8_t*>(view.data) + first;
#pragma unroll
      for (uint32_t k = 0; k < PK_ROUNDS; ++k) raw[k] = base[k * 64];
    } else {
      const uint32_t* base = static_cast<const uint32_t*>(view.data) + first;
#pragma unroll
      for (uint32_t k = 0; k < PK_ROUNDS; ++k) raw[k] = base[k * 64];
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
      const uint32_t r = wave_first + k * 64 + lane;
      const uint32_t index = view.row_begin + (r < row_count ? r : 0);
      raw[k] = view.kind == VIEW_FOR16 ? static_cast<const uint16_t*>(view.data)[index] : view.kind == VIEW_FOR8 ? static_cast<const uint8_t*>(view.data)[index] : static_cast<const uint32_t*>(view.data)[index];
    }
  }
  const uint32_t bias = view.kind == VIEW_INT32 ? 0u : static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[(view.row_begin + (wave_first < row_count ? wave_first : 0u)) / HY_FOR_BLOCK_SIZE]);
  const uint32_t origin = static_cast<uint32_t>(a.rank.key_min), range = static_cast<uint32_t>(a.rank.range);
  const uint32_t mask = a.radix_bits ? (1u << a.radix_bits) - 1 : 0u;
  u32x2_t entry[PK_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    raw[k] += bias;                                  // the key's low 32 bits
    const uint32_t distance = raw[k] - origin;
    entry[k] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(a.rank.entries) + (distance <= range ? (distance >> 5) * 8u : 0u));
  }
  uint32_t valid = 0, found = 0;
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    const uint32_t distance = raw[k] - origin;
    const uint32_t bit = distance & 31;
    const bool in = wave_first + k * 64 + lane < row_count;
    if (in) valid |= 1u << k;
    if (in && distance <= range && ((entry[k].x >> bit) & 1)) found |= 1u << k;
    rank[k] = entry[k].y + __popc(entry[k].x & ((1u << bit) - 1));
  }
  if (a.build_bloom && !a.keep_nulls && __any((valid & ~found) != 0)) {   // join_hash_steps.hpp:354-358
#pragma unroll
    for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
      if (((valid & ~found) >> k) & 1) { if (a.build_bloom[raw[k] & (BLOOM_BITS - 1)] == 0) valid &= ~(1u << k); }
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    bool null_partner;
    const bool emit = pk_emits<INNER>(a.mode, (found >> k) & 1, &null_partner);
    meta[k] = (valid >> k) & 1 ? (raw[k] & mask) | (null_partner ? 0x200u : 0u) | (emit ? 0x400u : 0u) : INVALID_PARTITION;
  }
}

// ---- pass 2 -------------------------------------------------------------------------------------------------------------------
// LDS, in 4-byte words: staged pairs {row | partition << 13 | null partner << 21, partner's rank} (one spare slot per
// partition, see rt_probe_emit) | pairs per (wave, partition), then the first slot of (wave, partition) | global pair index of
// staging slot 0 per partition | wave totals of the partition scan, reserved slots.
constexpr uint32_t PK_STAGE_ROW = 0x1FFF, PK_STAGE_PARTITION_SHIFT = 13, PK_STAGE_NULL = 1u << 21;
__host__ __device__ constexpr size_t pk_emit_lds_words(uint32_t partitions) {
  return 2 * (size_t{PK_TILE} + partitions + 2) + size_t{PK_WAVES} * partitions + partitions + 16;
}

template <int BUILD, bool PLAIN>
__device__ __forceinline__ void pk_copy_out(const PkArgs& a, const u32x2_t* s_stage, const uint32_t* s_out_base, uint32_t reserved, uint32_t chunk, uint32_t tile_row_begin, uint32_t tid) {
  u32x2_t* probe_out = reinterpret_cast<u32x2_t*>(a.probe_out);
  u32x2_t* build_out = reinterpret_cast<u32x2_t*>(a.build_out);
  auto rank_row = [&](uint32_t r) -> u32x2_t {
    if constexpr (BUILD == BUILD_IDENTITY_65535) {
      uint32_t c = r >> 16, offset = (r & 0xFFFFu) + c;   // r = c * 65535 + (c + low): offset < 2^17
      if (offset >= 65535u) { ++c; offset -= 65535u; }
      if (offset >= 65535u) { ++c; offset -= 65535u; }
      return u32x2_t{c, offset};
    } else if constexpr (BUILD == BUILD_IDENTITY) {
      uint32_t c = static_cast<uint32_t>(static_cast<double>(r) * a.rank.identity_inverse);
      if (c * a.rank.identity_rows > r) --c;
      uint32_t offset = r - c * a.rank.identity_rows;
      if (offset >= a.rank.identity_rows) { ++c; offset -= a.rank.identity_rows; }
      return u32x2_t{c, offset};
    } else if constexpr (BUILD == BUILD_PACKED) {
      const uint32_t id = a.ids32[r];
      return u32x2_t{id >> 16, id & 0xFFFFu};
    } else {
      return reinterpret_cast<const u32x2_t*>(a.row_ids)[r];
    }
  };
  auto store4 = [&](u32x4_t v, u32x2_t* at) {
    if constexpr (PLAIN) *reinterpret_cast<u32x4_t*>(at) = v; else __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(at));
  };
  auto store2 = [&](u32x2_t v, u32x2_t* at) {
    if constexpr (PLAIN) *at = v; else __builtin_nontemporal_store(v, at);
  };
  for (uint32_t slot = 2 * tid; slot < reserved; slot += 2 * PK_THREADS) {
    const u32x4_t records = *reinterpret_cast<const u32x4_t*>(s_stage + slot);
    const uint32_t tag0 = records.x, tag1 = slot + 1 < reserved ? records.z : STAGE_INVALID;
    const bool valid0 = tag0 != STAGE_INVALID, valid1 = tag1 != STAGE_INVALID;
    const uint32_t partition0 = (tag0 >> PK_STAGE_PARTITION_SHIFT) & 0xFF, partition1 = (tag1 >> PK_STAGE_PARTITION_SHIFT) & 0xFF;
    const u32x2_t probe0 = {chunk, tile_row_begin + (tag0 & PK_STAGE_ROW)}, probe1 = {chunk, tile_row_begin + (tag1 & PK_STAGE_ROW)};
    u32x2_t build0 = {0xFFFFFFFFu, 0xFFFFFFFFu}, build1 = {0xFFFFFFFFu, 0xFFFFFFFFu};
    if constexpr (BUILD != BUILD_NONE) {
      if (valid0 && !(tag0 & PK_STAGE_NULL)) build0 = rank_row(records.y);
      if (valid1 && !(tag1 & PK_STAGE_NULL)) build1 = rank_row(records.w);
    }
    if (valid0 && valid1 && partition0 == partition1) {   // both pairs of one run: its first global index has the slot's parity -> aligned
      const size_t pair_pos = static_cast<uint32_t>(s_out_base[partition0] + slot);
      store4(u32x4_t{probe0.x, probe0.y, probe1.x, probe1.y}, probe_out + pair_pos);
      if constexpr (BUILD != BUILD_NONE) store4(u32x4_t{build0.x, build0.y, build1.x, build1.y}, build_out + pair_pos);
    } else {
      if (valid0) {
        const size_t pair_pos = static_cast<uint32_t>(s_out_base[partition0] + slot);
        store2(probe0, probe_out + pair_pos);
        if constexpr (BUILD != BUILD_NONE) store2(build0, build_out + pair_pos);
      }
      if (valid1) {
        const size_t pair_pos = static_cast<uint32_t>(s_out_base[partition1] + slot + 1);
        store2(probe1, probe_out + pair_pos);
        if constexpr (BUILD != BUILD_NONE) store2(build1, build_out + pair_pos);
      }
    }
  }
}

template <bool INNER>
__global__ __launch_bounds__(PK_THREADS, 4) void pk_emit(PkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t join_smem[];
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t stage_slots = PK_TILE + partitions + 2;
  u32x2_t* s_stage = reinterpret_cast<u32x2_t*>(join_smem);                      // [stage_slots]
  uint32_t* s_wave_pairs = reinterpret_cast<uint32_t*>(s_stage + stage_slots);   // [PK_WAVES][partitions]
  uint32_t* s_out_base = s_wave_pairs + PK_WAVES * partitions;                   // [partitions]
  uint32_t* s_scratch = s_out_base + partitions;                                 // [4] wave totals of the partition scan, [8] reserved slots
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t scan_waves = partitions > 64 ? partitions / 64 : 1;
  const uint32_t tile = pk_block_tile(a.n_tiles);
  if (tile >= a.n_tiles || !a.plan->fits) return;
  const SliceView view = a.views[tile];
  if (view.row_count == 0) return;
  for (uint32_t i = tid; i < PK_WAVES * partitions; i += PK_THREADS) s_wave_pairs[i] = 0;
  // thread = partition: the cell's pairs and its first global pair index (fits 32 bits: pk_path_applies)
  const size_t cell = static_cast<size_t>(tid < partitions ? tid : 0) * a.stride + tile;
  const uint32_t cell_pairs = tid < partitions ? a.counts[cell] >> 16 : 0;
  const uint32_t cell_base = static_cast<uint32_t>(a.origin_pairs[tid < partitions ? tid : 0]) + a.rel_pairs[cell];
  uint32_t meta[PK_ROUNDS], rank[PK_ROUNDS];
  pk_evaluate<INNER>(a, view, wave, lane, meta, rank);
  // (a) reserve pairs + 1 slots per non-empty partition: scan inside each wave now, across waves in (c)
  const uint32_t reserve = cell_pairs ? cell_pairs + 1 : 0;
  uint32_t first_in_wave = 0;
  if (wave < scan_waves) {
    const uint32_t inclusive = join_wave_inclusive_scan(reserve);
    first_in_wave = inclusive - reserve;
    if (lane == 63) s_scratch[wave] = inclusive;
  }
  __syncthreads();   // the counters are zero
  // (b) rank inside the wave: ONE returning LDS atomic per pair.  The LDS serves the lanes of one instruction that hit the same
  // counter in lane order and a wave's LDS instructions in program order (lds_atomic_order_probe checks it once per process; the
  // host takes the general kernels where it does not hold), so a lane gets back the pairs of its partition in lower lanes and
  // earlier rounds.  The rank moves into meta[k] bits 11..
  {
    uint32_t before[PK_ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
      before[k] = 0;
      if (meta[k] & 0x400u) before[k] = atomicAdd(&s_wave_pairs[wave * partitions + (meta[k] & 0xFF)], 1u);
    }
#pragma unroll
    for (uint32_t k = 0; k < PK_ROUNDS; ++k) meta[k] |= before[k] << 11;
  }
  __syncthreads();
  // (c) thread = partition: first slot of every (wave, partition) = first slot of the partition (parity of its first global pair
  // index) + pairs of earlier waves; output base
  if (tid < partitions) {
    uint32_t first = first_in_wave + (wave > 0 ? s_scratch[0] : 0u) + (wave > 1 ? s_scratch[1] : 0u) + (wave > 2 ? s_scratch[2] : 0u);
    if (reserve) {
      const uint32_t shift = (first ^ cell_base) & 1u;
      s_stage[shift ? first : first + cell_pairs].x = STAGE_INVALID;   // the spare slot
      first += shift;
    }
    s_out_base[tid] = cell_base - first;
    uint32_t run = first;
#pragma unroll
    for (uint32_t w = 0; w < PK_WAVES; ++w) {
      const uint32_t pairs = s_wave_pairs[w * partitions + tid];
      s_wave_pairs[w * partitions + tid] = run;
      run += pairs;
    }
    if (tid == 0) s_scratch[8] = s_scratch[0] + (scan_waves > 1 ? s_scratch[1] : 0u) + (scan_waves > 2 ? s_scratch[2] : 0u) + (scan_waves > 3 ? s_scratch[3] : 0u);   // every reserved slot
  }
  __syncthreads();
  // (d) stage
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    if (!(meta[k] & 0x400u)) continue;
    const uint32_t partition = meta[k] & 0xFF;
    const uint32_t slot = s_wave_pairs[wave * partitions + partition] + (meta[k] >> 11);
    const uint32_t r = wave * PK_WAVE_ROWS + k * 64 + lane;
    s_stage[slot] = u32x2_t{r | (partition << PK_STAGE_PARTITION_SHIFT) | ((meta[k] & 0x200u) ? PK_STAGE_NULL : 0u), rank[k]};
  }
  __syncthreads();
  // (e) copy out: one loop per way of turning a partner's rank into its RowID (the identity cases have no global load in the loop:
  // no `s_waitcnt vmcnt(0)` per iteration, which would also wait for every store in flight)
  const uint32_t reserved = s_scratch[8];
  const uint32_t chunk = view.chunk, row_begin = view.row_begin;
#define HY_PK_COPY(BUILD)                                                                                        \
  do {                                                                                                           \
    if (a.plain_stores) pk_copy_out<BUILD, true>(a, s_stage, s_out_base, reserved, chunk, row_begin, tid);    \
    else pk_copy_out<BUILD, false>(a, s_stage, s_out_base, reserved, chunk, row_begin, tid);                  \
  } while (0)
  if (!a.build_out) HY_PK_COPY(BUILD_NONE);
  else if (a.rank.identity_rows == 65535u) HY_PK_COPY(BUILD_IDENTITY_65535);
  else if (a.rank.identity_rows) HY_PK_COPY(BUILD_IDENTITY);
  else if (a.ids32) HY_PK_COPY(BUILD_PACKED);
  else HY_PK_COPY(BUILD_ROW_IDS);
#undef HY_PK_COPY
}

// ---- the 131 070-element cuts ------------------------------------------------------------------------------------------------
// One workgroup per output PosList: its group (partition or probe chunk), the tile that holds the PosList's first element
// (64-ary searches over the scanned counts), then the tile's rows once more.
__global__ __launch_bounds__(PK_THREADS) void pk_cuts(PkArgs a) {
  __shared__ uint32_t s_members[PK_WAVES * PK_ROUNDS], s_emitters[PK_WAVES * PK_ROUNDS];
  const uint32_t slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (!a.plan->fits || slice >= a.plan->n_slices) return;
  uint32_t lo = 0, hi = a.n_groups;   // last group whose first PosList is <= slice (every wave searches: the results are uniform)
  while (hi - lo > 1) {
    const uint32_t step = (hi - lo + 63) / 64, at = lo + lane * step;
    const uint32_t below = __popcll(__ballot(at < hi && a.slice_base[at] <= slice));
    lo += (below - 1) * step;
    hi = lo + step < hi ? lo + step : hi;
  }
  const uint32_t group = lo;
  const uint32_t partition = a.radix_bits ? group : 0;
  const uint32_t* rel_elements = a.rel_elements + static_cast<size_t>(partition) * a.stride;
  const uint32_t first_tile = a.radix_bits ? 0 : a.group_first_tile[group], end_tile = a.radix_bits ? a.n_tiles : a.group_first_tile[group + 1];
  const uint32_t target = rel_elements[first_tile] + (slice - a.slice_base[group]) * PROBE_SIZE_PER_CHUNK;
  uint32_t tile = first_tile, tile_end = end_tile;   // last tile of the group whose first element is <= target: it holds the element
  while (tile_end - tile > 1) {
    const uint32_t step = (tile_end - tile + 63) / 64, at = tile + lane * step;
    const uint32_t below = __popcll(__ballot(at < tile_end && rel_elements[at] <= target));
    tile += (below - 1) * step;
    tile_end = tile + step < tile_end ? tile + step : tile_end;
  }
  const uint32_t cut_rank = target - rel_elements[tile];
  const SliceView view = a.views[tile];
  uint32_t meta[PK_ROUNDS], rank[PK_ROUNDS];
  pk_evaluate<false>(a, view, wave, lane, meta, rank);
  uint64_t members[PK_ROUNDS], emitters[PK_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    const bool member = (meta[k] & 0x1FF) == partition;   // (INVALID_PARTITION is no partition)
    members[k] = __ballot(member);
    emitters[k] = __ballot(member && (meta[k] & 0x400u));
    if (lane == 0) { s_members[wave * PK_ROUNDS + k] = __popcll(members[k]); s_emitters[wave * PK_ROUNDS + k] = __popcll(emitters[k]); }
  }
  __syncthreads();
  uint32_t members_before = 0, emitters_before = 0;   // rows come wave by wave, round by round, lane by lane
  for (uint32_t i = 0; i < wave * PK_ROUNDS; ++i) { members_before += s_members[i]; emitters_before += s_emitters[i]; }
  const uint64_t lower_lanes = (1ull << lane) - 1;
  const uint64_t cell_base = a.origin_pairs[partition] + a.rel_pairs[static_cast<size_t>(partition) * a.stride + tile];
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    const bool member = (members[k] >> lane) & 1;
    if (member && members_before + __popcll(members[k] & lower_lanes) == cut_rank) a.slice_offsets[slice] = cell_base + emitters_before + __popcll(emitters[k] & lower_lanes);
    members_before += __popcll(members[k]);
    emitters_before += __popcll(emitters[k]);
  }
}
