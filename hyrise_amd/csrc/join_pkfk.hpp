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

#ifndef HY_PK_TILE
#define HY_PK_TILE 8192
#endif
#ifndef HY_PK_WAVES_PER_SIMD
#define HY_PK_WAVES_PER_SIMD 4   // pk_emit: workgroups per CU x waves per workgroup / 4 (the register budget: 512 / this)
#endif
constexpr uint32_t PK_TILE = HY_PK_TILE;                 // rows per probe tile: a slice (8192) or half a slice (-DHY_PK_TILE=4096: A/B builds)
constexpr uint32_t PK_TILES_PER_SLICE = SLICE_ROWS / PK_TILE;
constexpr uint32_t PK_ROUNDS = 16;                       // rows per lane
constexpr uint32_t PK_THREADS = PK_TILE / PK_ROUNDS;     // 512
constexpr uint32_t PK_WAVES = PK_THREADS / 64;           // 8
constexpr uint32_t PK_WAVE_ROWS = PK_TILE / PK_WAVES;    // 1024 consecutive rows per wave
constexpr uint32_t PK_COUNT_WAVE_ROWS = 2048;            // pk_count: one FrameOfReference block per wave
constexpr uint32_t PK_COUNT_THREADS = 64 * (PK_TILE / PK_COUNT_WAVE_ROWS);   // 256
constexpr uint32_t PK_COUNT_BATCHES = PK_COUNT_WAVE_ROWS / 512;              // 4 batches of 512 rows, eight consecutive rows per lane
constexpr uint32_t PK_SCAN_THREADS = 512;
constexpr uint32_t PK_SCAN_CHUNK = 8192;                 // tiles per pass of pk_scan's loop (config 3's 7 323 tiles: one pass)
static_assert(SLICE_ROWS % PK_TILE == 0 && HY_FOR_BLOCK_SIZE % PK_WAVE_ROWS == 0 && HY_FOR_BLOCK_SIZE == PK_COUNT_WAVE_ROWS && PK_TILE % PK_COUNT_WAVE_ROWS == 0,
              "a wave's rows must lie in one FrameOfReference block");
static_assert(PK_TILE <= (1u << 13), "a staged pair keeps its row in 13 bits");

struct PkArgs {
  const SliceView* views;          // the probe column's slices; a tile is a slice or (PK_TILES_PER_SLICE == 2) half of one
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
  hy_row_id* build_out;            // nullptr: Semi / Anti
  hy_row_id* probe_out;
  uint64_t* slice_offsets;
  uint32_t emit_group_shift;       // pk_emit: log2 of the consecutive tiles an XCD takes (32: every XCD its own eighth of the tiles)
  uint32_t* row_ranks;             // probe keys without locality (Inner): pass 1 leaves every row's partner rank here (PK_NO_RANK: none), pass 2 reads it back
  uint32_t cut_blocks;             // pk_emit: its first cut_blocks workgroups compute the PosList cuts (pk_cut_slice)
  uint32_t bloom_is_bits;          // build_bloom holds 2^20 bits (a hinted build, rank_table_fill_checked), not one byte per bit
  uint64_t* trace;                 // debug (HY_JOIN_TRACE): 6 wall-clock stamps per pk_emit tile, else nullptr
  // The build side staged in LDS (pk_count_lds): a rank table of at most PK_LDS_KEYS key values -- a filtered dimension of a star join --
  // is 128 KB of presence bits; pass 1 looks every probe row up THERE and leaves two bits per row behind for pass 2.
  uint8_t* row_masks;              // [n_tiles][2][PK_TILE / 8] found | materialised, or nullptr (the classic kernels)
  // A build side filled on the strength of the column's key hint (rank_table_fill_checked): pk_plan confirms the fill's verdict
  // against the hint -- a column that contradicts it gets the plan "does not fit": nothing is written, the host runs the join again.
  const BuildVerdict* verdict;     // rank_table_fill_checked's; nullptr: the build side was not hinted, or ...
  const uint64_t* fill_records;    // ... rank_table_fill_waves filled it: [n_fill_records][4] smallest key | largest key (both ^ sign) | flags, one per workgroup
  uint32_t n_fill_records;
  uint64_t hint_min, hint_max;
  uint32_t hint_allows_duplicates;
  hy_join_status* status;          // HY_JOIN_ASYNC: what the host would read from the mailbox, in device memory; else nullptr
  FillCleaning cleaning;           // pk_emit: two regions its first workgroups zero (the rank table and filter of the join BEFORE: ZeroedBlocks in join.hip)
};
constexpr uint32_t PK_LDS_KEYS = 1u << 20;                       // the table's range must be smaller: then the Bloom filter (bit = key & 0xFFFFF) is the presence bit
constexpr uint32_t PK_LDS_WORDS = PK_LDS_KEYS / 32;              // 32 768 presence words = 128 KB
constexpr uint32_t PK_LDS_SUBTILES = 4;                          // tiles a 1024-thread workgroup of pk_count_lds counts at once
constexpr uint32_t PK_LDS_MAX_PARTITIONS = 128;                 // (the workgroup's cells: 4 tiles x partitions x 8 copies -- 16 KB beside the 128 KB of bits)
__host__ __device__ inline uint32_t pk_lds_presence_words(uint64_t range) { return (static_cast<uint32_t>(range >> 5) + 1 + 3) & ~3u; }   // (16-byte aligned cells behind them)
__host__ __device__ inline size_t pk_count_lds_bytes(uint64_t range, uint32_t partitions) { return 4 * (size_t{pk_lds_presence_words(range)} + size_t{PK_LDS_SUBTILES} * partitions * COUNT_COPIES); }

// The rows of tile `tile`: its slice's view, narrowed to the tile.
__device__ __forceinline__ SliceView pk_tile_view(const PkArgs& a, uint32_t tile) {
  SliceView view = uniform_view(a.views + tile / PK_TILES_PER_SLICE);   // (tile: the same for every lane of the workgroup)
  if constexpr (PK_TILES_PER_SLICE > 1) {
    const uint32_t offset = (tile % PK_TILES_PER_SLICE) * PK_TILE;
    view.row_begin += offset;
    view.row_count = view.row_count > offset ? (view.row_count - offset < PK_TILE ? view.row_count - offset : PK_TILE) : 0;
  }
  return view;
}

// Is the bit of `key` set in the build side's Bloom filter (join_hash_steps.hpp:354-358: index = hash & 0xFFFFF, the hash of an integer
// key is the key)?
__device__ __forceinline__ bool pk_bloom_test(const PkArgs& a, uint32_t key) {
  const uint32_t index = key & (BLOOM_BITS - 1);
  if (a.bloom_is_bits) return (reinterpret_cast<const uint32_t*>(a.build_bloom)[index >> 5] >> (index & 31)) & 1;
  return a.build_bloom[index] != 0;
}

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
// One workgroup per tile; a wave owns 2048 rows of it (one FrameOfReference block): four batches of 512 rows, a lane holds eight
// consecutive rows of each (16-byte loads).  Where the 50 us go (measured by switching parts off): 43 are the tile's words arriving --
// a plain read of the column in this launch shape takes 23 (tools/hbm_read.hip); a workgroup spends two dependent round trips, the
// tile's view and then its words, of its short life waiting -- lookups and LDS atomics are the other 7.  Persistent workgroups that
// request their next tile's words ahead were SLOWER (76 us): loads return in order (one counter, vmcnt), so either the lookups queue
// behind the prefetch or the prefetch costs 48 more live registers.
// LDS: the presence words come from `presence` (LDS, the whole table), partner-less rows ask the same words instead of the Bloom
// filter (exact for a range below 2^20: the one key value of the range that shares the row's filter bit), and the rows' found /
// materialised bits are left in `masks` (the tile's 2 x 1 KB) for pass 2.
// RANKS: probe keys without locality (shuffled foreign keys against a table of megabytes): every lookup is a sector out of the L2 or the
// memory-side cache, and pass 2 would make every one of them again.  Pass 1 then reads the whole entry, computes the partner's rank and
// leaves it in `tile_ranks` (4 bytes a row, written and read back sequentially): 60 M shuffled probe rows 0.81 + 1.02 ms -> see DESIGN.md 4.2.
constexpr uint32_t PK_NO_RANK = 0xFFFFFFFFu;
template <uint32_t WIDTH, bool LDS = false, bool RANKS = false>
__device__ __forceinline__ void pk_count_wave(const PkArgs& a, const SliceView& view, uint32_t wave, uint32_t lane, uint32_t* cells, const uint32_t* presence = nullptr,
                                              uint8_t* masks = nullptr, uint32_t* tile_ranks = nullptr) {
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
  // (the wave's block minimum: one value for all lanes, through the scalar cache -- immutable like the views)
  typedef __attribute__((address_space(4))) const uint32_t constant_word;
  const uint32_t bias = view.kind == VIEW_INT32 ? 0u : ((constant_word*)view.aux)[__builtin_amdgcn_readfirstlane(static_cast<int>((view.row_begin + wave_first) / HY_FOR_BLOCK_SIZE))];
  const uint32_t origin = static_cast<uint32_t>(a.rank.key_min), range = static_cast<uint32_t>(a.rank.range);
  const uint32_t mask = a.radix_bits ? (1u << a.radix_bits) - 1 : 0u;
  const uint32_t copy = ((lane >> 4) & 3) | ((lane & 1) << 2);   // (lanes 16 apart -- 32 orders, the period of dbgen's sparse keys mod 128 -- meet in one partition: they use different copies)
  uint32_t counted = 0;   // (uniform) bit b: batch b was counted by the lean loop
  // The lean loop -- Inner / Semi joins, a wave whose 2048 rows all exist, the table in global memory: about a dozen vector instructions
  // per row (the general loop below spends twice that on masks for rows that do not exist and on merging runs of equal partitions, and
  // is bound by it: 38 of pk_count's 47 us at SF10 were instruction issue).  A key outside the table's range reads the empty entry behind
  // the table (distance clamped to range + 32: every build path allocates and clears that entry), so a row is one unpack, one add, one
  // clamp, a shift and a mask for the address, one bit test -- and ONE LDS atomic, no run detection: with eight copies per cell the
  // lanes of an instruction rarely meet.  A batch in which some row has no partner goes to the general loop (the Bloom filter decides
  // about such rows), as does every other mode.
  if constexpr (!LDS && !RANKS) {
    if ((a.mode == HY_JOIN_INNER || a.mode == HY_JOIN_SEMI) && wave_first + PK_COUNT_WAVE_ROWS <= row_count) {
      const uint32_t delta = bias - origin, beyond = range + 32;
      const char* table = reinterpret_cast<const char*>(a.rank.entries);
      uint32_t counted_here = 0;
#pragma unroll
      for (uint32_t g = 0; g < PK_COUNT_BATCHES; g += 2) {
        uint32_t bits[2][8];
#pragma unroll
        for (uint32_t b = 0; b < 2; ++b) {
#pragma unroll
          for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t distance = batch_word<WIDTH>(words[g + b], j) + delta;
            const uint32_t clamped = distance < beyond ? distance : beyond;
            bits[b][j] = *reinterpret_cast<const uint32_t*>(table + ((clamped >> 2) & ~7u));
          }
        }
#pragma unroll
        for (uint32_t b = 0; b < 2; ++b) {
          uint32_t found = 0;
#pragma unroll
          for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t distance = batch_word<WIDTH>(words[g + b], j) + delta;
            found |= (distance <= range ? (bits[b][j] >> (distance & 31)) & 1u : 0u) << j;
          }
          if (__any(found != 0xFFu)) continue;   // (uniform: the batch is counted by the general loop)
#pragma unroll
          for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t partition = (batch_word<WIDTH>(words[g + b], j) + bias) & mask;
            atomicAdd(&cells[partition * COUNT_COPIES + copy], 0x10001u);
          }
          counted_here |= 1u << (g + b);
        }
      }
      counted = counted_here;
      if (counted == (1u << PK_COUNT_BATCHES) - 1) return;
    }
  }
  // Two groups of 1024 rows: a group's sixteen lookups per lane are in flight together (counting only needs an entry's presence bits,
  // its first word), and the kernel stays at 64 registers -- eight workgroups per CU hide each other's round trips.
#pragma unroll
  for (uint32_t g = 0; g < PK_COUNT_BATCHES; g += 2) {
    uint32_t bits[2][8];
    uint32_t bases[RANKS ? 2 : 1][RANKS ? 8 : 1];
#pragma unroll
    for (uint32_t b = 0; b < 2; ++b) {
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t distance = batch_word<WIDTH>(words[g + b], j) + bias - origin;   // (32-bit: both sides' keys are int32 values, pk_path in run_join)
        if constexpr (LDS) bits[b][j] = presence[distance <= range ? distance >> 5 : 0u];
        else if constexpr (RANKS) {
          const u32x2_t entry = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(a.rank.entries) + (distance <= range ? (distance >> 5) * 8u : 0u));
          bits[b][j] = entry.x;
          bases[b][j] = entry.y;
        } else bits[b][j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.rank.entries) + (distance <= range ? (distance >> 5) * 8u : 0u));
      }
    }
    if constexpr (RANKS) {   // the lane's eight consecutive rows of both batches: 32 bytes each, two 16-byte stores
#pragma unroll
      for (uint32_t b = 0; b < 2; ++b) {
        uint32_t ranks[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t distance = batch_word<WIDTH>(words[g + b], j) + bias - origin;
          const bool has = distance <= range && ((bits[b][j] >> (distance & 31)) & 1);
          ranks[j] = has ? bases[b][j] + __popc(bits[b][j] & ((1u << (distance & 31)) - 1)) : PK_NO_RANK;
        }
        const uint32_t first = wave_first + (g + b) * 512 + lane * 8;
        if (first < row_count) {   // (a partial tile's last lanes write ranks of rows that do not exist: inside the tile's PK_TILE words)
          u32x4_t* out = reinterpret_cast<u32x4_t*>(tile_ranks + first);
          out[0] = u32x4_t{ranks[0], ranks[1], ranks[2], ranks[3]};
          out[1] = u32x4_t{ranks[4], ranks[5], ranks[6], ranks[7]};
        }
      }
    }
#pragma unroll
    for (uint32_t b = 0; b < 2; ++b) {
      if ((counted >> (g + b)) & 1) continue;
      uint32_t valid = 0, found = 0;
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t distance = batch_word<WIDTH>(words[g + b], j) + bias - origin;
        const bool in = wave_first + (g + b) * 512 + lane * 8 + j < row_count;
        valid |= (in ? 1u : 0u) << j;
        found |= (in && distance <= range && ((bits[b][j] >> (distance & 31)) & 1) ? 1u : 0u) << j;
      }
      if (a.build_bloom && !a.keep_nulls && __any((valid & ~found) != 0)) {   // partner-less rows: materialised only if the build side's filter has their bit
        uint32_t miss = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          if (!(((valid & ~found) >> j) & 1)) continue;
          bool filter_bit;
          if constexpr (LDS) {   // range < 2^20: the filter's bit of a key is the presence bit of the one key of the range congruent to it
            const uint32_t candidate = (batch_word<WIDTH>(words[g + b], j) + bias - origin) & (BLOOM_BITS - 1);
            filter_bit = candidate <= range && ((presence[candidate >> 5] >> (candidate & 31)) & 1);
          } else {
            filter_bit = pk_bloom_test(a, batch_word<WIDTH>(words[g + b], j) + bias);
          }
          miss |= (filter_bit ? 0u : 1u) << j;
        }
        valid &= ~miss;
      }
      if constexpr (LDS) {   // the lane's eight rows are consecutive: one byte each of the tile's found / materialised bits
        const uint32_t at = (wave_first + (g + b) * 512) / 8 + lane;
        masks[at] = static_cast<uint8_t>(found);
        masks[PK_TILE / 8 + at] = static_cast<uint8_t>(valid);
      }
      // neighbouring rows share their key (four lineitems per order): the lane's eight consecutive rows leave as one LDS atomic per RUN
      // of equal partitions (same-address atomics serialise)
      uint32_t run_partition = 0xFFFFFFFFu, run_value = 0;
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        if (!((valid >> j) & 1)) continue;
        bool null_partner;
        const bool emit = pk_emits<false>(a.mode, (found >> j) & 1, &null_partner);
        const uint32_t partition = (batch_word<WIDTH>(words[g + b], j) + bias) & mask, value = emit ? 0x10001u : 1u;
        if (partition == run_partition) { run_value += value; continue; }
        if (run_value) atomicAdd(&cells[run_partition * COUNT_COPIES + copy], run_value);
        run_partition = partition;
        run_value = value;
      }
      if (run_value) atomicAdd(&cells[run_partition * COUNT_COPIES + copy], run_value);
    }
  }
}

#ifndef HY_PK_COUNT_WGS_PER_CU
#define HY_PK_COUNT_WGS_PER_CU 8   // pk_count<false>: resident workgroups per CU the register budget is cut for (A/B builds: 5, 6)
#endif
template <bool RANKS = false>
__global__ __launch_bounds__(PK_COUNT_THREADS, (RANKS ? 4 : HY_PK_COUNT_WGS_PER_CU) * PK_COUNT_THREADS / 256) void pk_count(PkArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cells[MAX_PARTITIONS * COUNT_COPIES];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t tile = pk_block_tile(a.n_tiles);
  if (blockIdx.x == 0 && tid == 0) *a.ticket = 0;   // pk_scan's arrival counter (pk_scan runs behind this kernel)
  if (tile >= a.n_tiles) return;
  const SliceView view = pk_tile_view(a, tile);
  for (uint32_t i = tid; i < partitions * COUNT_COPIES; i += PK_COUNT_THREADS) s_cells[i] = 0;
  __syncthreads();
  uint32_t* tile_ranks = RANKS ? a.row_ranks + static_cast<size_t>(tile) * PK_TILE : nullptr;
  if (view.kind == VIEW_FOR8) pk_count_wave<1, false, RANKS>(a, view, wave, lane, s_cells, nullptr, nullptr, tile_ranks);
  else if (view.kind == VIEW_FOR16) pk_count_wave<2, false, RANKS>(a, view, wave, lane, s_cells, nullptr, nullptr, tile_ranks);
  else pk_count_wave<4, false, RANKS>(a, view, wave, lane, s_cells, nullptr, nullptr, tile_ranks);
  __syncthreads();
  for (uint32_t partition = tid; partition < partitions; partition += PK_COUNT_THREADS) {
    const u32x4_t low = *reinterpret_cast<const u32x4_t*>(s_cells + partition * COUNT_COPIES), high = *reinterpret_cast<const u32x4_t*>(s_cells + partition * COUNT_COPIES + 4);
    a.counts[static_cast<size_t>(partition) * a.stride + tile] = low.x + low.y + low.z + low.w + high.x + high.y + high.z + high.w;
  }
}

// Pass 1 with the build side staged in LDS -- what north_star asks of JoinHash ("build side staged in LDS", per-wave probing) where it
// fits: a rank table over fewer than 2^20 key values (the filtered dimension of a star join: SSB's part has 1 M keys) is 128 KB of
// presence bits.  A probe row of the classic kernel pulls a 128-byte line of the table out of the L2 for one bit -- with unclustered
// foreign keys (SSB lineorder) that line traffic, not HBM, bounds both passes (pk_count 1.0 ms for 180 M rows).  Here every CU stages
// the bits once, persistent 1024-thread workgroups walk the tiles four at a time, and pass 2 gets the rows' two bits from a mask
// instead of looking non-partners up again (pk_emit<INNER, true>).
__global__ __launch_bounds__(1024) void pk_count_lds(PkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t pk_lds[];
  uint32_t* s_presence = pk_lds;                                        // [range / 32 + 1]
  uint32_t* s_cells = pk_lds + pk_lds_presence_words(a.rank.range);     // [PK_LDS_SUBTILES][partitions * COUNT_COPIES]
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t sub = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), local = tid & 255;
  const uint32_t partitions = 1u << a.radix_bits;
  if (blockIdx.x == 0 && tid == 0) *a.ticket = 0;   // pk_scan's arrival counter (pk_scan runs behind this kernel)
  const uint32_t words = static_cast<uint32_t>(a.rank.range >> 5) + 1;
  for (uint32_t i = tid; i < words; i += 1024) s_presence[i] = a.rank.entries[i].x;
  uint32_t* cells = s_cells + sub * partitions * COUNT_COPIES;
  for (uint32_t first = blockIdx.x * PK_LDS_SUBTILES; first < a.n_tiles; first += gridDim.x * PK_LDS_SUBTILES) {
    const uint32_t tile = first + sub;
    for (uint32_t i = local; i < partitions * COUNT_COPIES; i += 256) cells[i] = 0;
    __syncthreads();   // (the bits are staged; the cells are zero)
    if (tile < a.n_tiles) {
      const SliceView view = pk_tile_view(a, tile);
      uint8_t* masks = a.row_masks + static_cast<size_t>(tile) * (2 * PK_TILE / 8);
      // (a partial tile: the bits of rows that do not exist are written as zeros by the waves that have rows; waves without rows clear theirs)
      if (wave * PK_COUNT_WAVE_ROWS >= view.row_count) {
        for (uint32_t i = lane; i < PK_COUNT_WAVE_ROWS / 8; i += 64) { masks[wave * PK_COUNT_WAVE_ROWS / 8 + i] = 0; masks[PK_TILE / 8 + wave * PK_COUNT_WAVE_ROWS / 8 + i] = 0; }
      } else if (view.kind == VIEW_FOR8) pk_count_wave<1, true>(a, view, wave, lane, cells, s_presence, masks);
      else if (view.kind == VIEW_FOR16) pk_count_wave<2, true>(a, view, wave, lane, cells, s_presence, masks);
      else pk_count_wave<4, true>(a, view, wave, lane, cells, s_presence, masks);
    }
    __syncthreads();
    if (tile < a.n_tiles) {
      for (uint32_t partition = local; partition < partitions; partition += 256) {
        const u32x4_t low = *reinterpret_cast<const u32x4_t*>(cells + partition * COUNT_COPIES), high = *reinterpret_cast<const u32x4_t*>(cells + partition * COUNT_COPIES + 4);
        a.counts[static_cast<size_t>(partition) * a.stride + tile] = low.x + low.y + low.z + low.w + high.x + high.y + high.z + high.w;
      }
    }
    __syncthreads();   // (the cells are read before the next round clears them)
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
    const uint64_t elements = tid < partitions ? __hip_atomic_load(a.totals + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const uint64_t pairs = tid < partitions ? __hip_atomic_load(a.totals + partitions + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
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
    n_pairs = __hip_atomic_load(a.totals + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (one workgroup: its own stores, the prefixes below included)
    uint64_t running = 0;
    for (uint32_t begin = 0; begin < a.n_groups; begin += PK_SCAN_THREADS) {
      const uint32_t g = begin + tid;
      uint64_t slices = 0;
      if (g < a.n_groups) {
        const uint32_t elements = a.rel_elements[a.group_first_tile[g + 1] * PK_TILES_PER_SLICE] - a.rel_elements[a.group_first_tile[g] * PK_TILES_PER_SLICE];
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
  // The records of rank_table_fill_waves' workgroups (written by an earlier kernel of this stream): extent and flags of the build column.
  uint64_t build_low = ~0ull, build_high = 0, build_flags = 0;
  if (a.fill_records) {
    for (uint32_t i = tid; i < a.n_fill_records; i += PK_SCAN_THREADS) {
      const uint64_t* record = a.fill_records + 4 * size_t{i};
      build_low = record[0] < build_low ? record[0] : build_low;
      build_high = record[1] > build_high ? record[1] : build_high;
      build_flags |= record[2];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      build_flags |= __shfl_xor(build_flags, d, 64);
      const uint64_t other_low = __shfl_xor(build_low, d, 64), other_high = __shfl_xor(build_high, d, 64);
      build_low = other_low < build_low ? other_low : build_low;
      build_high = other_high > build_high ? other_high : build_high;
    }
    __syncthreads();   // (s_tmp may still be read from the scans above)
    if ((tid & 63) == 0) { s_tmp[tid >> 6] = build_low; s_tmp[8 + (tid >> 6)] = build_high; s_tmp[16 + (tid >> 6)] = build_flags; }
    __syncthreads();
  }
  if (tid == 0) {
    bool confirmed = true;
    if (a.fill_records) {
      for (uint32_t v = 0; v < PK_SCAN_THREADS / 64; ++v) {
        build_low = s_tmp[v] < build_low ? s_tmp[v] : build_low;
        build_high = s_tmp[8 + v] > build_high ? s_tmp[8 + v] : build_high;
        build_flags |= s_tmp[16 + v];
      }
      constexpr uint64_t SIGN = 1ull << 63;
      confirmed = !(build_flags & 1) && (!(build_flags & 2) || a.hint_allows_duplicates) && !(build_flags & 4) && (build_low ^ SIGN) == a.hint_min && (build_high ^ SIGN) == a.hint_max;
    } else if (a.verdict) {   // (written by an earlier kernel of this stream)
      const BuildVerdict v = *a.verdict;
      confirmed = v.done && !v.unsorted_signed && (!v.equal_neighbours || a.hint_allows_duplicates) && !v.outside_hint && v.key_min == a.hint_min && v.key_max == a.hint_max;
    }
    const uint32_t fits = confirmed && n_pairs <= a.capacity && n_slices <= a.slice_capacity ? 1u : 0u;
    a.plan->fits = fits;
    a.plan->n_slices = n_slices;
    if (fits && a.slice_offsets) a.slice_offsets[n_slices] = n_pairs;
    if (a.status) {
      a.status->n_pairs = n_pairs;
      a.status->n_slices = n_slices;
      a.status->fits = fits;
      a.status->build_confirmed = confirmed ? 1u : 0u;
      a.status->error = 0;
      a.status->reserved = 0;
    } else {   // (a synchronous join: the host reads the pinned mailbox when the stream has drained; HY_JOIN_ASYNC never looks at it -- no
               //  writes across the host link and no system-scope fence at the end of every asynchronous join's plan)
      a.mailbox->n_pairs = n_pairs;
      a.mailbox->n_slices = n_slices;
      a.mailbox->n_uncached = 0;
      a.mailbox->fits = fits;
      a.mailbox->build_unconfirmed = confirmed ? 0u : 1u;
      __threadfence_system();
    }
  }
}

__global__ __launch_bounds__(PK_SCAN_THREADS) void pk_scan(PkArgs a) {
  // element i of a chunk sits at i + i / 16: a thread's 16 consecutive elements start 17 words apart -> no bank conflicts
  constexpr uint32_t PER_THREAD = PK_SCAN_CHUNK / PK_SCAN_THREADS, PADDED = PK_SCAN_CHUNK + PK_SCAN_CHUNK / PER_THREAD;
  __shared__ uint32_t s_elements[PADDED], s_pairs[PADDED];
  __shared__ uint64_t s_tmp[24];   // (scans: 8 wave totals; pk_plan's reduction of the build side's records: 3 x 8)
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
  // The workgroup that arrives last plans the output.  It needs the partitions' totals (and, without radix partitioning, its own
  // row of prefixes): they leave with agent-scope atomic stores (write-through) and are read back with agent-scope atomic loads --
  // no release fence, which would write back the whole L2.
  if (tid == 0) {
    rel_elements[a.n_tiles] = static_cast<uint32_t>(carry);
    rel_pairs[a.n_tiles] = static_cast<uint32_t>(carry >> 32);
    __hip_atomic_store(a.totals + partition, static_cast<uint32_t>(carry), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.totals + gridDim.x + partition, static_cast<uint32_t>(carry >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t arrived = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = arrived + 1 == gridDim.x ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  pk_plan(a, s_tmp, tid);
}

// The stored words of a wave's 1024 rows, requested with 16-byte loads: lane l asks for the words of rows [16 l, 16 l + 16) of the
// wave's block -- one piece of 16 bytes for 1-byte offsets, two for 2-byte offsets, four for 4-byte words -- and pk_rows_of_words
// hands every lane ITS rows (row k * 64 + lane in round k: the order the ranking needs) through a wave-private LDS block.  Two
// coalesced loads per lane instead of sixteen 2-byte ones, and a prefetched tile costs eight to sixteen registers.
// A piece is loaded if its first row exists (a partial tile's last piece reads < 16 bytes past the rows: inside the segment's padding).
struct PkWords { u32x4_t piece[4]; };
__device__ __forceinline__ void pk_load_words(const SliceView& view, uint32_t wave, uint32_t lane, PkWords& words) {
  const uint32_t width = view.kind == VIEW_FOR8 ? 1u : view.kind == VIEW_FOR16 ? 2u : 4u;
  const uint32_t first = wave * PK_WAVE_ROWS + lane * PK_ROUNDS;          // of the lane's 16 rows, in the tile
  const char* base = static_cast<const char*>(view.data) + static_cast<size_t>(view.row_begin + first) * width;
  const uint32_t rows_per_piece = 16 / width;
#pragma unroll
  for (uint32_t p = 0; p < 4; ++p) {
    words.piece[p] = u32x4_t{0, 0, 0, 0};
    // (global, not flat: a flat load also counts on lgkmcnt)
    typedef const __attribute__((address_space(1))) u32x4_t* global_words;
    if (p < width && first + p * rows_per_piece < view.row_count) words.piece[p] = *(global_words)(base + p * 16);
  }
}

// The lane's words go into the wave's LDS block (`block`: 4 KB that belong to this wave for the moment) ...
__device__ __forceinline__ void pk_words_to_block(const SliceView& view, const PkWords& words, uint32_t* block, uint32_t lane) {
  const uint32_t width = view.kind == VIEW_FOR8 ? 1u : view.kind == VIEW_FOR16 ? 2u : 4u;
  u32x4_t* mine = reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(block) + lane * PK_ROUNDS * width);
#pragma unroll
  for (uint32_t p = 0; p < 4; ++p) {
    if (p < width) mine[p] = words.piece[p];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ... and come back as the stored words of the lane's rows: row k * 64 + lane of the wave's block in round k.
__device__ __forceinline__ uint32_t pk_block_word(const SliceView& view, const uint32_t* block, uint32_t lane, uint32_t k) {
  if (view.kind == VIEW_FOR16) return reinterpret_cast<const uint16_t*>(block)[k * 64 + lane];
  if (view.kind == VIEW_FOR8) return reinterpret_cast<const uint8_t*>(block)[k * 64 + lane];
  return block[k * 64 + lane];
}

// Keys, rank-table entries, the rows' fate -- HALVES == 2: eight rows at a time (two table round trips instead of one, half the
// registers at the peak).
// meta[k] = partition | null_partner << 9 | emit << 10 (INVALID_PARTITION: the row is not materialised); rank[k] = the partner's rank.
// MASKS: pass 1 was pk_count_lds -- a row's found / materialised bits come from `tile_masks` (the tile's 2 x PK_TILE / 8 bytes), only rows
// with a partner read their table entry (the others read entry 0: one line for all of them), the Bloom filter is not asked again.
// RANKS (Inner joins): pass 1 was pk_count<true> -- a row's rank (or PK_NO_RANK) comes from `tile_ranks`, no table entry is read; a row
// without a partner emits nothing in an Inner join, whether it counted as materialised is pass 1's business.
template <bool INNER, uint32_t HALVES, bool MASKS = false, bool RANKS = false>
__device__ __forceinline__ void pk_lookup_rows(const PkArgs& a, const SliceView& view, const uint32_t* block, uint32_t wave, uint32_t lane, uint32_t (&meta)[PK_ROUNDS],
                                               uint32_t (&rank)[PK_ROUNDS], const uint8_t* tile_masks = nullptr, const uint32_t* tile_ranks = nullptr) {
  constexpr uint32_t N = PK_ROUNDS / HALVES;
  const uint32_t row_count = view.row_count;
  const uint32_t wave_first = wave * PK_WAVE_ROWS;
  const uint32_t bias = view.kind == VIEW_INT32 ? 0u : static_cast<uint32_t>(static_cast<const int32_t*>(view.aux)[(view.row_begin + (wave_first < row_count ? wave_first : 0u)) / HY_FOR_BLOCK_SIZE]);
  const uint32_t origin = static_cast<uint32_t>(a.rank.key_min), range = static_cast<uint32_t>(a.rank.range);
  const uint32_t mask = a.radix_bits ? (1u << a.radix_bits) - 1 : 0u;
  // MASKS: the wave's 1024 rows are 32 words of found bits and 32 of materialised bits; lane l < 32 holds found word l, lane 32 + l
  // materialised word l; row k * 64 + lane sits in word 2 k + lane / 32, bit lane % 32
  uint32_t mask_word = 0;
  if constexpr (MASKS) mask_word = reinterpret_cast<const uint32_t*>(tile_masks + (lane < 32 ? 0u : PK_TILE / 8))[wave * (PK_WAVE_ROWS / 32) + (lane & 31)];
#pragma unroll
  for (uint32_t h = 0; h < HALVES; ++h) {
    uint32_t raw[N];
    u32x2_t entry[N];
    uint32_t valid = 0, found = 0;
    if constexpr (RANKS) {
      static_assert(INNER && !MASKS, "ranks are handed over by Inner joins on the global-table kernels");
#pragma unroll
      for (uint32_t j = 0; j < N; ++j) {
        const uint32_t r = wave_first + (h * N + j) * 64 + lane;
        raw[j] = pk_block_word(view, block, lane, h * N + j) + bias;
        rank[h * N + j] = r < row_count ? tile_ranks[r] : PK_NO_RANK;
      }
#pragma unroll
      for (uint32_t j = 0; j < N; ++j) {
        if (wave_first + (h * N + j) * 64 + lane < row_count) valid |= 1u << j;
        if (rank[h * N + j] != PK_NO_RANK) found |= 1u << j;
      }
    } else if constexpr (MASKS) {
#pragma unroll
      for (uint32_t j = 0; j < N; ++j) {
        const uint32_t k = h * N + j;
        const uint32_t found_word = lane < 32 ? __builtin_amdgcn_readlane(mask_word, 2 * k) : __builtin_amdgcn_readlane(mask_word, 2 * k + 1);
        const uint32_t valid_word = lane < 32 ? __builtin_amdgcn_readlane(mask_word, 32 + 2 * k) : __builtin_amdgcn_readlane(mask_word, 32 + 2 * k + 1);
        found |= ((found_word >> (lane & 31)) & 1u) << j;
        valid |= ((valid_word >> (lane & 31)) & 1u) << j;
      }
#pragma unroll
      for (uint32_t j = 0; j < N; ++j) {
        raw[j] = pk_block_word(view, block, lane, h * N + j) + bias;
        const uint32_t distance = raw[j] - origin;
        entry[j] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(a.rank.entries) + (((found >> j) & 1) ? (distance >> 5) * 8u : 0u));
      }
#pragma unroll
      for (uint32_t j = 0; j < N; ++j) {
        const uint32_t bit = (raw[j] - origin) & 31;
        rank[h * N + j] = entry[j].y + __popc(entry[j].x & ((1u << bit) - 1));
      }
    } else {
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
      raw[j] = pk_block_word(view, block, lane, h * N + j) + bias;   // the key's low 32 bits
      const uint32_t distance = raw[j] - origin;                     // (32-bit: both sides' keys are int32 values, pk_path in run_join)
      entry[j] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(a.rank.entries) + (distance <= range ? (distance >> 5) * 8u : 0u));
    }
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
      const uint32_t distance = raw[j] - origin;
      const uint32_t bit = distance & 31;
      const bool in = wave_first + (h * N + j) * 64 + lane < row_count;
      if (in) valid |= 1u << j;
      if (in && distance <= range && ((entry[j].x >> bit) & 1)) found |= 1u << j;
      rank[h * N + j] = entry[j].y + __popc(entry[j].x & ((1u << bit) - 1));
    }
    if (a.build_bloom && !a.keep_nulls && __any((valid & ~found) != 0)) {   // join_hash_steps.hpp:354-358
#pragma unroll
      for (uint32_t j = 0; j < N; ++j) {
        if (((valid & ~found) >> j) & 1) { if (!pk_bloom_test(a, raw[j])) valid &= ~(1u << j); }
      }
    }
    }
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
      bool null_partner;
      const bool emit = pk_emits<INNER>(a.mode, (found >> j) & 1, &null_partner);
      meta[h * N + j] = (valid >> j) & 1 ? (raw[j] & mask) | (null_partner ? 0x200u : 0u) | (emit ? 0x400u : 0u) : INVALID_PARTITION;
    }
    if (h + 1 < HALVES) __builtin_amdgcn_sched_barrier(0);   // (the second half's loads must not move up: that is the point)
  }
}

// ---- evaluation of a tile's rows (cuts) --------------------------------------------------------------------------------------
// Wave w owns rows [1024 w, 1024 (w + 1)) of the tile, row k * 64 + lane in round k: row order = (wave, round, lane).
template <bool INNER, bool MASKS = false>
__device__ __forceinline__ void pk_evaluate(const PkArgs& a, const SliceView& view, uint32_t* block, uint32_t wave, uint32_t lane, uint32_t (&meta)[PK_ROUNDS], uint32_t (&rank)[PK_ROUNDS],
                                            const uint8_t* tile_masks = nullptr) {
  PkWords words;
  pk_load_words(view, wave, lane, words);
  pk_words_to_block(view, words, block, lane);
  pk_lookup_rows<INNER, 1, MASKS>(a, view, block, wave, lane, meta, rank, tile_masks);
}

// ---- pass 2 -------------------------------------------------------------------------------------------------------------------
// LDS, in 4-byte words: staged pairs {row | partition << 13 | null partner << 21, partner's rank} (one spare slot per
// partition, see rt_probe_emit) | pairs per (wave, partition), then the first slot of (wave, partition) | global pair index of
// staging slot 0 per partition | wave totals of the partition scan, reserved slots.
constexpr uint32_t PK_STAGE_ROW = 0x1FFF, PK_STAGE_PARTITION_SHIFT = 13, PK_STAGE_NULL = 1u << 21;
__host__ __device__ constexpr size_t pk_emit_lds_words(uint32_t partitions) {
  return 2 * (size_t{PK_TILE} + partitions + 2) + size_t{PK_WAVES} * partitions + 3 * size_t{partitions} + 32;
}

// STORES: 0 nontemporal | 1 write-back | 2 write-back for the two pairs that share a 128-byte line with a NEIGHBOURING tile's pairs (a
// run's first and last line: the partial lines merge in the XCD's L2 instead of reaching HBM as two masked writes each), nontemporal
// for the full lines in between (nothing of the 0.96 GB stays behind in the L2 for the next kernel to evict).  tools/hbm_write.hip:
// 246 us / 187 us / 188 us for config 3's layout.  s_edge: [2][partitions] first and last line of every run.
template <int BUILD, int STORES>
__device__ __forceinline__ void pk_copy_out(const PkArgs& a, const u32x2_t* s_stage, const uint32_t* s_out_base, const uint32_t* s_edge, uint32_t partitions, uint32_t reserved,
                                            uint32_t chunk, uint32_t tile_row_begin, uint32_t tid) {
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
  auto store2 = [&](u32x2_t v, u32x2_t* at) {   // (a run's first or last pair)
    if constexpr (STORES != 0) *at = v; else __builtin_nontemporal_store(v, at);
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
      const uint32_t pair_index = s_out_base[partition0] + slot;
      const size_t pair_pos = pair_index;
      const u32x4_t probe_pairs = {probe0.x, probe0.y, probe1.x, probe1.y}, build_pairs = {build0.x, build0.y, build1.x, build1.y};
      bool write_back = STORES == 1;
      if constexpr (STORES == 2) write_back = (pair_index >> 4) == s_edge[partition0] || (pair_index >> 4) == s_edge[partitions + partition0];
      if (write_back) {
        *reinterpret_cast<u32x4_t*>(probe_out + pair_pos) = probe_pairs;
        if constexpr (BUILD != BUILD_NONE) *reinterpret_cast<u32x4_t*>(build_out + pair_pos) = build_pairs;
      } else {
        __builtin_nontemporal_store(probe_pairs, reinterpret_cast<u32x4_t*>(probe_out + pair_pos));
        if constexpr (BUILD != BUILD_NONE) __builtin_nontemporal_store(build_pairs, reinterpret_cast<u32x4_t*>(build_out + pair_pos));
      }
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

// One tile from its stored words to its pairs in the output.  The five phases are separated by four workgroup barriers.
template <bool INNER, bool MASKS = false, bool RANKS = false>
__device__ __forceinline__ void pk_emit_tile(const PkArgs& a, uint32_t tile, const SliceView& view, const PkWords& words, uint32_t cell_pairs, uint32_t cell_base,
                                             uint32_t* join_smem, uint32_t tid, uint32_t lane, uint32_t wave) {
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t stage_slots = PK_TILE + partitions + 2;
  u32x2_t* s_stage = reinterpret_cast<u32x2_t*>(join_smem);                      // [stage_slots]
  uint32_t* s_wave_pairs = reinterpret_cast<uint32_t*>(s_stage + stage_slots);   // [PK_WAVES][partitions]
  uint32_t* s_out_base = s_wave_pairs + PK_WAVES * partitions;                   // [partitions]
  uint32_t* s_edge = s_out_base + partitions;                                    // [2][partitions] first / last 128-byte line of the partition's run
  uint32_t* s_scratch = s_edge + 2 * partitions;                                 // [4] wave totals of the partition scan, [8] reserved slots, [16 + wave] a counter for rows without a pair
  const uint32_t scan_waves = partitions > 64 ? partitions / 64 : 1;
  if (a.trace && tid == 0) a.trace[tile * 6 + 0] = wall_clock64();
  for (uint32_t i = tid; i < PK_WAVES * partitions; i += PK_THREADS) s_wave_pairs[i] = 0;
  uint32_t meta[PK_ROUNDS], rank[PK_ROUNDS];
  pk_words_to_block(view, words, join_smem + wave * 1024, lane);   // (the staging area is not in use yet: 4 KB of it per wave)
  pk_lookup_rows<INNER, 1, MASKS, RANKS>(a, view, join_smem + wave * 1024, wave, lane, meta, rank, MASKS ? a.row_masks + static_cast<size_t>(tile) * (2 * PK_TILE / 8) : nullptr,
                                         RANKS ? a.row_ranks + static_cast<size_t>(tile) * PK_TILE : nullptr);
  __builtin_amdgcn_wave_barrier();
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
  // (b) rank inside the wave: ONE returning LDS atomic per pair.  The LDS serves the lanes of one instruction that hit the same
  // counter in lane order and a wave's LDS instructions in program order (lds_atomic_order_probe checks it once per process; the
  // host takes the general kernels where it does not hold), so a lane gets back the pairs of its partition in lower lanes and
  // earlier rounds.  The rank moves into meta[k] bits 11..
  {
    // (rows without a pair count on a spare counter of the wave: sixteen atomics back to back, no branch around any of them)
    uint32_t before[PK_ROUNDS];
    uint32_t* mine = s_wave_pairs + wave * partitions;
    uint32_t* spare = s_scratch + 16 + wave;
    if constexpr (MASKS) {   // (a selective join: most rows have no pair -- on one spare counter their atomics would serialise, 64 lanes deep)
#pragma unroll
      for (uint32_t k = 0; k < PK_ROUNDS; ++k) { before[k] = 0; if (meta[k] & 0x400u) before[k] = atomicAdd(mine + (meta[k] & 0xFF), 1u); }
      (void)spare;
    } else {
#pragma unroll
    for (uint32_t k = 0; k < PK_ROUNDS; ++k) before[k] = atomicAdd((meta[k] & 0x400u) ? mine + (meta[k] & 0xFF) : spare, 1u);
    }
#pragma unroll
    for (uint32_t k = 0; k < PK_ROUNDS; ++k) meta[k] |= (meta[k] & 0x400u) ? before[k] << 11 : 0u;
  }
  __syncthreads();
  if (a.trace && tid == 0) a.trace[tile * 6 + 2] = wall_clock64();
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
    s_edge[tid] = cell_base >> 4;
    s_edge[partitions + tid] = (cell_base + cell_pairs - (cell_pairs ? 1u : 0u)) >> 4;
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
  if (a.trace && tid == 0) a.trace[tile * 6 + 3] = wall_clock64();
  // (d) stage
  // (rows without a pair write the slot behind every run: no branch here either)
#pragma unroll
  for (uint32_t k = 0; k < PK_ROUNDS; ++k) {
    const uint32_t partition = meta[k] & 0xFF;
    const uint32_t slot = (meta[k] & 0x400u) ? s_wave_pairs[wave * partitions + partition] + (meta[k] >> 11) : stage_slots - 1;
    const uint32_t r = wave * PK_WAVE_ROWS + k * 64 + lane;
    s_stage[slot] = u32x2_t{r | (partition << PK_STAGE_PARTITION_SHIFT) | ((meta[k] & 0x200u) ? PK_STAGE_NULL : 0u), rank[k]};
  }
  __syncthreads();
  if (a.trace && tid == 0) a.trace[tile * 6 + 4] = wall_clock64();
  // (e) copy out: one loop per way of turning a partner's rank into its RowID (the identity cases have no global load in the loop:
  // no `s_waitcnt vmcnt(0)` per iteration, which would also wait for every store in flight)
  const uint32_t reserved = s_scratch[8];
  const uint32_t chunk = view.chunk, row_begin = view.row_begin;
  // (STORES = 2: the flavour the A/Bs of rounds 3-5 kept -- all nontemporal 356 us, all write-back 294 us but 25 us more in the kernels behind it,
  //  mixed 302 us, profiles/r04_join_variants.txt)
#define HY_PK_COPY(BUILD) pk_copy_out<BUILD, static_cast<int>(FIXED_JOIN_STORES)>(a, s_stage, s_out_base, s_edge, partitions, reserved, chunk, row_begin, tid)
  if (!a.build_out) HY_PK_COPY(BUILD_NONE);
  else if (a.rank.identity_rows == 65535u) HY_PK_COPY(BUILD_IDENTITY_65535);
  else if (a.rank.identity_rows) HY_PK_COPY(BUILD_IDENTITY);
  else if (a.ids32) HY_PK_COPY(BUILD_PACKED);
  else HY_PK_COPY(BUILD_ROW_IDS);
#undef HY_PK_COPY
  if (a.trace && tid == 0) a.trace[tile * 6 + 5] = wall_clock64();
}

// ---- the 131 070-element cuts ------------------------------------------------------------------------------------------------
// One workgroup per output PosList: its group (partition or probe chunk), the tile that holds the PosList's first element
// (64-ary searches over the scanned counts), then the tile's rows once more.
// smem: PK_WAVES * 1024 + 2 * PK_WAVES * PK_ROUNDS words.
template <bool MASKS = false>
__device__ __forceinline__ void pk_cut_slice(const PkArgs& a, uint32_t slice, uint32_t* smem, uint32_t tid, uint32_t lane, uint32_t wave) {
  uint32_t* s_rows = smem;
  uint32_t* s_members = smem + PK_WAVES * 1024;
  uint32_t* s_emitters = s_members + PK_WAVES * PK_ROUNDS;
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
  const uint32_t first_tile = a.radix_bits ? 0 : a.group_first_tile[group] * PK_TILES_PER_SLICE, end_tile = a.radix_bits ? a.n_tiles : a.group_first_tile[group + 1] * PK_TILES_PER_SLICE;
  const uint32_t target = rel_elements[first_tile] + (slice - a.slice_base[group]) * PROBE_SIZE_PER_CHUNK;
  uint32_t tile = first_tile, tile_end = end_tile;   // last tile of the group whose first element is <= target: it holds the element
  while (tile_end - tile > 1) {
    const uint32_t step = (tile_end - tile + 63) / 64, at = tile + lane * step;
    const uint32_t below = __popcll(__ballot(at < tile_end && rel_elements[at] <= target));
    tile += (below - 1) * step;
    tile_end = tile + step < tile_end ? tile + step : tile_end;
  }
  const uint32_t cut_rank = target - rel_elements[tile];
  const SliceView view = pk_tile_view(a, tile);
  uint32_t meta[PK_ROUNDS], rank[PK_ROUNDS];
  pk_evaluate<false, MASKS>(a, view, s_rows + wave * 1024, wave, lane, meta, rank, MASKS ? a.row_masks + static_cast<size_t>(tile) * (2 * PK_TILE / 8) : nullptr);
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

__global__ __launch_bounds__(PK_THREADS) void pk_cuts(PkArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cut[PK_WAVES * 1024 + 2 * PK_WAVES * PK_ROUNDS];
  const uint32_t tid = threadIdx.x;
  pk_cut_slice(a, blockIdx.x, s_cut, tid, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6));
}

// One tile per workgroup (2 workgroups per CU overlap each other's phases).
template <bool INNER, bool MASKS = false, bool RANKS = false>
__global__ __launch_bounds__(PK_THREADS, HY_PK_WAVES_PER_SIMD) void pk_emit(PkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t join_smem[];
  const uint32_t partitions = 1u << a.radix_bits;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The first cut_blocks workgroups (a multiple of 8: the others keep their XCD) find the PosList cuts while the rest emits: a launch
  // of its own behind this kernel was 13 us of an otherwise idle device.
  for (uint32_t i = blockIdx.x * PK_THREADS + tid; i < a.cleaning.first_vectors + a.cleaning.second_vectors; i += gridDim.x * PK_THREADS) {
    if (i < a.cleaning.first_vectors) a.cleaning.first[i] = u32x4_t{0, 0, 0, 0}; else a.cleaning.second[i - a.cleaning.first_vectors] = u32x4_t{0, 0, 0, 0};
  }
  if (blockIdx.x < a.cut_blocks) { pk_cut_slice<MASKS>(a, blockIdx.x, join_smem, tid, lane, wave); return; }
  const uint32_t block = blockIdx.x - a.cut_blocks;
  // Which tile: the device works on one front of tiles that moves through the probe side -- workgroups arrive at the XCDs in turn (block b
  // on XCD b % 8), an XCD takes 2^emit_group_shift consecutive tiles of every 8 x 2^emit_group_shift (the partial lines two neighbouring
  // tiles share meet in one L2).  One front per XCD (every XCD its own eighth of the tiles: emit_group_shift = 32) is eight times as many
  // places written at the same time: the same write pattern alone runs 209 us that way and 187 us on one front (tools/write_fronts.hip),
  // pk_emit itself 265 .. 285 us instead of 285 .. 300 in the allocations where it is slow.
  uint32_t tile;
  if (a.emit_group_shift < 32) {
    const uint32_t xcd = block & 7, j = block >> 3, shift = a.emit_group_shift;
    const uint32_t whole = ((a.n_tiles + 7) / 8) >> shift << shift;   // (the grid: 8 x ceil(n_tiles / 8) tile blocks)
    tile = j < whole ? ((j >> shift) << (shift + 3)) + (xcd << shift) + (j & ((1u << shift) - 1)) : whole * 8 + (j - whole) * 8 + xcd;
  } else tile = (block & 7) * ((a.n_tiles + 7) / 8) + (block >> 3);
  if (tile >= a.n_tiles) return;
  const SliceView view = pk_tile_view(a, tile);
  const uint32_t fits = a.plan->fits;   // (asked for together with the view: one round trip, not two, before the words can be requested)
  if (!fits || view.row_count == 0) return;
  // thread = partition: the cell's pairs and its first global pair index (fits 32 bits: pk_path in run_join)
  const size_t cell = static_cast<size_t>(tid < partitions ? tid : 0) * a.stride + tile;
  const uint32_t cell_pairs = tid < partitions ? a.counts[cell] >> 16 : 0;
  const uint32_t cell_base = static_cast<uint32_t>(a.origin_pairs[tid < partitions ? tid : 0]) + a.rel_pairs[cell];
  PkWords words;
  pk_load_words(view, wave, lane, words);
  pk_emit_tile<INNER, MASKS, RANKS>(a, tile, view, words, cell_pairs, cell_base, join_smem, tid, lane, wave);
}

