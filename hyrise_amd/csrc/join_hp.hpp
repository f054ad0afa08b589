// join_hp.hpp -- the radix-partitioned JoinHash (included by join.hip behind join_pkfk.hpp, inside namespace hy).
//
// What it replaces (reference, CPU): materialize_input -> partition_by_radix -> build -> probe, operators/join_hash/join_hash_steps.hpp:274-420
// (materialise), :509-617 (partition_by_radix: low radix_bits bits of the key's hash, stable), :426-507 (build: one hash table per partition),
// :624-922 (probe, partition by partition), with the output order of :541-591, :655-760.  This is the path BASELINE.json's north_star words as
// "radix partition -> per-partition table staged in LDS -> per-wave probing".
//
// When it runs: unique int32 build keys that are not too sparse (the rank-table test of prepare_build) but NOT in a shape the primary-key /
// foreign-key kernels serve well -- a build column that is not sorted (join_pkfk.hpp would need its rank table filled with one random atomic
// per key and its RowIDs scattered: 0.56 + 0.41 ms for 15 M keys), or probe keys without locality (every probe row of pk_count / pk_emit then
// pulls its own 64-byte sector of the 15 MB table out of the memory-side cache: 0.81 + 1.01 ms for 60 M rows).  Both columns int32 values /
// FrameOfReference segments without NULLs (SliceViews), no secondary predicates; everything else keeps the general kernels.
//
// The pipeline -- every step a streaming pass, no host round trip between them:
//   hp_hist / scan / hp_scatter   both inputs become (key, RowID) tuples, partition by partition (partition = key & (2^radix_bits - 1), the
//                                 reference's radix partition), in row order inside a partition: stable, like partition_by_radix -- so the
//                                 probe tuples already lie in OUTPUT order.  8192-row tiles, ranks from one returning LDS atomic per row
//                                 (lane-ordered, join.hip lds_atomic_order_probe), runs staged in LDS and written contiguously
//   hp_layout                     partition boundaries and the 4096-tuple probe steps of every partition (a step never straddles partitions)
//   hp_table                      ONE workgroup per partition builds the partition's table IN LDS: a key of partition p is slot
//                                 (key - origin) >> radix_bits of p -- 32 slots per entry {presence bits, rank of the entry's first key}, at
//                                 most 120 KB -- one LDS atomic per key (a bit that is already set: duplicate keys, the join falls back),
//                                 ranks by a running sum of population counts, RowIDs by rank written next to the partition's tuples.  The
//                                 finished entries go to global memory, partition-major (15 MB for SF10 orders)
//   hp_probe                      persistent workgroups, one per CU: stage the partition's entries in LDS (one coalesced 117 KB read), then
//                                 step through its probe tuples: a lookup is ONE LDS read; per tuple a result word (materialised | emits | has
//                                 a partner | the partner's rank) -- 4 bytes, sequential -- and per step the materialised elements and pairs
//   hp_plan                       one workgroup: running sums over the steps (pair index of every step, element index inside the partition),
//                                 PosLists per partition (a new one every 131 070 materialised elements), capacity check, mailbox / status
//   hp_emit                       one workgroup per step: result words + RowIDs in, pairs out -- consecutive tuples write consecutive pairs,
//                                 so both PosLists are written in long sequential runs; the partner's RowID is one read of the partition's
//                                 rank -> RowID slice (L2-resident); a tuple that begins a PosList stores its pair index in slice_offsets
// HBM traffic at SF10 (15 M x 60 M): build 60 + 2 x 120, probe 2 x 240 + 480, table 2 x 15 + 60, probe pass 480 + 240, emit 240 + 240 + 960 MB.
#pragma once

constexpr uint32_t HP_TILE = SLICE_ROWS;                  // rows of a column per partitioning workgroup: a slice
constexpr uint32_t HP_THREADS = 512;
constexpr uint32_t HP_WAVES = HP_THREADS / 64;
constexpr uint32_t HP_ROUNDS = HP_TILE / HP_THREADS;      // rows per lane: row wave * 1024 + k * 64 + lane in round k
constexpr uint32_t HP_WAVE_ROWS = HP_TILE / HP_WAVES;
constexpr uint32_t HP_STEP = 4096;                        // probe tuples per step
constexpr uint32_t HP_PROBE_THREADS = 1024;
constexpr uint32_t HP_PER_THREAD = HP_STEP / HP_PROBE_THREADS;
constexpr uint32_t HP_MAX_TABLE_WORDS = 15360;            // entries of a partition's table that fit LDS: 120 KB
constexpr uint32_t HP_MATERIALISED = 1u << 31, HP_EMITS = 1u << 30, HP_FOUND = 1u << 29, HP_RANK = HP_FOUND - 1;   // result word of a probe tuple

struct HpSide {   // one input column as tuples, partition by partition
  const SliceView* views;
  uint32_t n_tiles;
  uint32_t stride;             // row stride of counts / bases (> n_tiles, the rest of a row is zero: bases[p * stride + n_tiles] = the partition's end)
  uint32_t radix_bits;
  uint32_t* counts;            // [P][stride] rows per (partition, tile)
  const uint64_t* bases;       // exclusive scan of counts (flat, partition-major)
  u32x2_t* tuples;             // [rows] {key (the int32 value's bits), chunk << 16 | chunk offset}
};

struct HpTable {
  u32x2_t* entries;            // [P][words] {presence bits of 32 slots, rank of the first of them inside the partition}
  uint32_t words;              // per partition
  uint32_t origin;             // a multiple of 2^radix_bits at or below the smallest key: slot = (key - origin) >> radix_bits
  uint32_t range;              // largest key - origin
  uint32_t radix_bits;
  uint32_t existence_only;     // Semi / Anti without secondary predicates: duplicate keys are fine, nobody asks for a rank
  uint32_t* ids;               // [build rows] RowID (chunk << 16 | offset) by partition and rank, or nullptr
  uint32_t* bloom_bits;        // the build side's Bloom filter (bit = key & 0xFFFFF, join_hash_steps.hpp:252,362), or nullptr -- partition by partition:
                               // partition p's keys can only set bits whose index has p in its low radix_bits bits, so p owns 2^(20 - radix_bits) of
                               // them: bit (p << (20 - radix_bits)) + ((key & 0xFFFFF) >> radix_bits) of this array
  uint32_t* flags;             // [0] a key twice
  uint32_t* partial_bits;      // [mark workgroups][words] hp_mark: the presence bits one workgroup found (hp_ranks combines a partition's)
  uint32_t* partial_bloom;     // [mark workgroups][bloom words per partition] likewise, or nullptr
};
__host__ __device__ inline uint32_t hp_bloom_words(uint32_t radix_bits) { return (BLOOM_BITS >> radix_bits) / 32 ? (BLOOM_BITS >> radix_bits) / 32 : 1u; }   // per partition
constexpr uint32_t HP_BLOOM_LDS_WORDS = 4096;   // a partition's filter slice is staged in LDS up to this size (radix_bits >= 3); else global atomics

struct HpLayout {              // written by hp_layout
  uint32_t* build_off;         // [P + 1] first build tuple of every partition
  uint32_t* probe_off;         // [G + 1] first probe tuple of every GROUP of the output: the partitions, or -- radix_bits == 0, one partition, the
                               //         tuples in row order -- the probe chunks (the reference then probes chunk by chunk: a PosList per chunk)
  uint32_t* first_step;        // [G + 1] first probe step; [G] = number of steps
  uint32_t n_groups;
  u32x4_t* steps;              // [steps][2] {first tuple, end of its group's tuples, group, partition} by hp_layout | {pairs of all earlier steps, materialised
                               //            elements of the group's earlier steps, first PosList of the group, first build tuple of the partition} by hp_plan
  uint32_t* mark_first;        // [P + 1] first workgroup of hp_mark / hp_ids that works on a partition (workgroups in proportion to its tuples)
  uint32_t mark_groups;        // workgroups hp_layout may hand out (P more exist: every partition with tuples gets at least one)
};

struct HpProbe {
  const u32x2_t* tuples;       // probe tuples, partition by partition
  HpLayout layout;
  HpTable table;
  uint32_t mode;
  uint32_t keep_nulls;
  uint32_t shares;             // workgroups per partition
  const uint32_t* bloom_bits;  // the build side's filter, or nullptr: every probe row counts as materialised
  uint32_t* results;           // [probe rows] HP_* | rank
  uint32_t* step_counts;       // [steps] materialised elements | pairs << 16
  // hp_plan / hp_emit
  uint32_t* pair_base;         // [steps] pairs of all earlier steps
  uint32_t* element_base;      // [steps] materialised elements of the partition's earlier steps
  uint32_t* slice_base;        // [G + 1] first output PosList of every group
  uint32_t* group_elements;    // [G] materialised elements of all earlier groups
  JoinPlan* plan;
  JoinMailbox* mailbox;
  hy_join_status* status;
  uint64_t capacity;
  uint32_t slice_capacity;
  hy_row_id* build_out;
  hy_row_id* probe_out;
  uint64_t* slice_offsets;
  uint32_t max_steps;
};

// ---- partitioning -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HP_THREADS) void hp_hist(HpSide s) {
  __shared__ uint32_t s_cells[MAX_PARTITIONS * COUNT_COPIES];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t partitions = 1u << s.radix_bits, mask = partitions - 1;
  const uint32_t tile = blockIdx.x;
  for (uint32_t i = tid; i < partitions * COUNT_COPIES; i += HP_THREADS) s_cells[i] = 0;
  __syncthreads();
  const SliceView view = s.views[tile];
  const uint32_t copy = lane & (COUNT_COPIES - 1);
#pragma unroll
  for (uint32_t k = 0; k < HP_ROUNDS; ++k) {
    const uint32_t r = wave * HP_WAVE_ROWS + k * 64 + lane;
    if (r < view.row_count) atomicAdd(&s_cells[(static_cast<uint32_t>(view_key(view, view.row_begin + r)) & mask) * COUNT_COPIES + copy], 1u);
  }
  __syncthreads();
  for (uint32_t p = tid; p < partitions; p += HP_THREADS) {
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t c = 0; c < COUNT_COPIES; ++c) sum += s_cells[p * COUNT_COPIES + c];
    s.counts[static_cast<size_t>(p) * s.stride + tile] = sum;
  }
}

// LDS, in 4-byte words: staged tuples [HP_TILE][2] | tuples per (wave, partition), then the first slot of (wave, partition) [HP_WAVES][P] |
// first staging slot of a partition [P] | its first tuple index minus that slot, 64 bits [P][2] | wave totals of the scan [8]
__host__ __device__ constexpr size_t hp_scatter_lds_words(uint32_t partitions) { return 2 * size_t{HP_TILE} + size_t{HP_WAVES} * partitions + 3 * size_t{partitions} + 2 + 8; }

__global__ __launch_bounds__(HP_THREADS) void hp_scatter(HpSide s) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hp_smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t partitions = 1u << s.radix_bits, mask = partitions - 1;
  u32x2_t* s_stage = reinterpret_cast<u32x2_t*>(hp_smem);
  uint32_t* s_wave = hp_smem + 2 * HP_TILE;                       // [HP_WAVES][partitions]
  uint32_t* s_first = s_wave + HP_WAVES * partitions;             // [partitions]
  uint64_t* s_base = reinterpret_cast<uint64_t*>(s_first + partitions + ((partitions & 1) ? 1 : 0));   // [partitions]
  uint32_t* s_totals = reinterpret_cast<uint32_t*>(s_base + partitions);   // [8]
  const uint32_t tile = blockIdx.x;
  const SliceView view = s.views[tile];
  if (view.row_count == 0) return;
  for (uint32_t i = tid; i < HP_WAVES * partitions; i += HP_THREADS) s_wave[i] = 0;
  uint32_t key[HP_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < HP_ROUNDS; ++k) {
    const uint32_t r = wave * HP_WAVE_ROWS + k * 64 + lane;
    key[k] = r < view.row_count ? static_cast<uint32_t>(view_key(view, view.row_begin + r)) : 0u;
  }
  __syncthreads();
  // rank inside (wave, partition): one returning LDS atomic per row -- the LDS serves the lanes of an instruction in lane order and a wave's
  // instructions in program order (checked once per process, lds_atomic_order_probe; the host does not take this path otherwise)
  uint32_t before[HP_ROUNDS];
#pragma unroll
  for (uint32_t k = 0; k < HP_ROUNDS; ++k) {
    const uint32_t r = wave * HP_WAVE_ROWS + k * 64 + lane;
    before[k] = 0;
    if (r < view.row_count) before[k] = atomicAdd(&s_wave[wave * partitions + (key[k] & mask)], 1u);
  }
  __syncthreads();
  // thread = partition: tuples of earlier waves, the partition's total; then the partitions' first staging slots (exclusive scan over partitions)
  uint32_t total = 0;
  if (tid < partitions) {
#pragma unroll
    for (uint32_t w = 0; w < HP_WAVES; ++w) {
      const uint32_t c = s_wave[w * partitions + tid];
      s_wave[w * partitions + tid] = total;
      total += c;
    }
  }
  const uint32_t inclusive = join_wave_inclusive_scan(total);
  if (lane == 63) s_totals[wave] = inclusive;
  __syncthreads();
  uint32_t earlier_waves = 0;
  for (uint32_t w = 0; w < wave; ++w) earlier_waves += s_totals[w];
  if (tid < partitions) {
    const uint32_t first = earlier_waves + inclusive - total;
    s_first[tid] = first;
    s_base[tid] = s.bases[static_cast<size_t>(tid) * s.stride + tile] - first;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < HP_ROUNDS; ++k) {
    const uint32_t r = wave * HP_WAVE_ROWS + k * 64 + lane;
    if (r >= view.row_count) continue;
    const uint32_t partition = key[k] & mask;
    s_stage[s_first[partition] + s_wave[wave * partitions + partition] + before[k]] = u32x2_t{key[k], (view.chunk << 16) | (view.row_begin + r)};
  }
  __syncthreads();
  for (uint32_t i = tid; i < view.row_count; i += HP_THREADS) {
    const u32x2_t tuple = s_stage[i];
    __builtin_nontemporal_store(tuple, s.tuples + (s_base[tuple.x & mask] + i));
  }
}

// Exclusive prefix over the HP_PROBE_THREADS threads of a workgroup (s_tmp: 16 words); *total = the sum.  Ends with a barrier.
__device__ __forceinline__ uint32_t hp_block_exclusive_scan(uint32_t v, uint32_t* s_tmp, uint32_t tid, uint32_t* total) {
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t inclusive = join_wave_inclusive_scan(v);
  __syncthreads();   // (s_tmp may still be read from an earlier call)
  if (lane == 63) s_tmp[wave] = inclusive;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < HP_PROBE_THREADS / 64; ++w) { const uint32_t t = s_tmp[w]; before += w < wave ? t : 0u; all += t; }
  *total = all;
  return before + inclusive - v;
}

// Partition boundaries of the build side; groups of the output, their probe steps, and the workgroups of hp_mark (one workgroup).
__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_layout(HpSide build, HpSide probe, const uint64_t* probe_row_base, HpLayout out) {
  __shared__ uint32_t s_tmp[16];
  const uint32_t tid = threadIdx.x, partitions = 1u << probe.radix_bits;
  const uint64_t build_rows = build.bases[static_cast<size_t>(partitions - 1) * build.stride + build.n_tiles];
  uint32_t groups_before = 0;   // hp_mark's workgroups: in proportion to the partition's tuples, at least one where there are tuples
  for (uint32_t begin = 0; begin < partitions; begin += HP_PROBE_THREADS) {
    const uint32_t p = begin + tid;
    uint32_t mine = 0;
    if (p < partitions) {
      const uint64_t first = build.bases[static_cast<size_t>(p) * build.stride], end = build.bases[static_cast<size_t>(p) * build.stride + build.n_tiles];
      out.build_off[p] = static_cast<uint32_t>(first);
      if (p + 1 == partitions) out.build_off[partitions] = static_cast<uint32_t>(end);
      mine = end > first ? static_cast<uint32_t>(((end - first) * out.mark_groups + build_rows - 1) / build_rows) : 0u;
    }
    uint32_t total = 0;
    const uint32_t before = groups_before + hp_block_exclusive_scan(mine, s_tmp, tid, &total);
    if (p < partitions) out.mark_first[p] = before;
    groups_before += total;
  }
  if (tid == 0) out.mark_first[partitions] = groups_before;
  uint32_t steps_before = 0;
  for (uint32_t begin = 0; begin < out.n_groups; begin += HP_PROBE_THREADS) {
    const uint32_t g = begin + tid;
    uint64_t first = 0, end = 0;
    if (g < out.n_groups) {
      if (probe.radix_bits) { first = probe.bases[static_cast<size_t>(g) * probe.stride]; end = probe.bases[static_cast<size_t>(g) * probe.stride + probe.n_tiles]; }
      else { first = probe_row_base[g]; end = probe_row_base[g + 1]; }
      out.probe_off[g] = static_cast<uint32_t>(first);
      if (g + 1 == out.n_groups) out.probe_off[out.n_groups] = static_cast<uint32_t>(end);
    }
    const uint32_t steps = static_cast<uint32_t>((end - first + HP_STEP - 1) / HP_STEP);
    uint32_t total = 0;
    const uint32_t before = steps_before + hp_block_exclusive_scan(steps, s_tmp, tid, &total);
    if (g < out.n_groups) {
      out.first_step[g] = before;
      for (uint32_t k = 0; k < steps; ++k)
        out.steps[2 * (before + k)] = u32x4_t{static_cast<uint32_t>(first) + k * HP_STEP, static_cast<uint32_t>(end), g, probe.radix_bits ? g : 0u};
    }
    steps_before += total;
  }
  if (tid == 0) out.first_step[out.n_groups] = steps_before;
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): logical workgroup (b % 8) * (grid / 8) + b / 8 gives XCD x the x-th
// eighth of the logical ids -- the workgroups of one partition (neighbours in logical order) meet in one L2, where its table entries, its
// rank -> RowID slice and the lines its RowIDs are scattered to stay.  (grid: a multiple of 8.)
__device__ __forceinline__ uint32_t hp_logical_block() { return (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3); }

// ---- the build side's tables ---------------------------------------------------------------------------------------------------------
// hp_mark: a workgroup takes a share of ONE partition's build tuples (hp_layout hands every partition workgroups in proportion to its
// tuples: dbgen's order keys populate 32 of 128 partitions) and sets their keys' presence bits in a table IN LDS -- one LDS atomic per key;
// 15 M device-scope atomics on the table itself took 0.9 ms -- and the bits they set in the partition's slice of the Bloom filter likewise;
// what it found goes to global memory as it is (58 KB, coalesced).  hp_ranks, one workgroup per partition, combines the shares' bits, turns
// population counts into ranks (fewer bits than tuples: a key twice) and writes the partition's entries and filter slice.  hp_ids writes
// every tuple's RowID to the partition's rank -> RowID slice.
__device__ __forceinline__ uint32_t hp_mark_partition(const HpLayout& layout, uint32_t partitions, uint32_t group) {
  uint32_t lo = 0, hi = partitions;   // the last partition whose first workgroup is <= group
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (layout.mark_first[mid] <= group) lo = mid; else hi = mid; }
  return lo;
}

// The tuples of hp_mark / hp_ids workgroup `group`: share `group - mark_first[partition]` of its partition.
__device__ __forceinline__ bool hp_mark_share(const HpLayout& layout, uint32_t partitions, uint32_t group, uint32_t* partition, uint32_t* begin, uint32_t* end) {
  if (group >= layout.mark_first[partitions]) return false;
  const uint32_t p = hp_mark_partition(layout, partitions, group);
  const uint32_t shares = layout.mark_first[p + 1] - layout.mark_first[p], share = group - layout.mark_first[p];
  const uint32_t first = layout.build_off[p], n = layout.build_off[p + 1] - first, per = (n + shares - 1) / shares;
  *partition = p;
  *begin = first + (share * per < n ? share * per : n);
  *end = first + ((share + 1) * per < n ? (share + 1) * per : n);
  return true;
}

__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_mark(HpSide build, HpLayout layout, HpTable t) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hp_smem[];
  const uint32_t tid = threadIdx.x, partitions = 1u << t.radix_bits;
  uint32_t partition, begin, end;
  const uint32_t block = hp_logical_block();
  if (!hp_mark_share(layout, partitions, block, &partition, &begin, &end)) return;
  uint32_t* s_bits = hp_smem;                       // [words]
  uint32_t* s_bloom = hp_smem + t.words;            // [bloom words of a partition] if staged
  const uint32_t bloom_words = hp_bloom_words(t.radix_bits);
  const bool bloom_in_lds = t.bloom_bits && bloom_words <= HP_BLOOM_LDS_WORDS;
  for (uint32_t i = tid; i < t.words; i += HP_PROBE_THREADS) s_bits[i] = 0;
  if (bloom_in_lds) for (uint32_t i = tid; i < bloom_words; i += HP_PROBE_THREADS) s_bloom[i] = 0;
  __syncthreads();
  for (uint32_t first = begin + tid; first < end; first += 4 * HP_PROBE_THREADS) {
    uint32_t keys[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) { const uint32_t i = first + j * HP_PROBE_THREADS; keys[j] = i < end ? build.tuples[i].x : 0u; }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t key = keys[j], distance = key - t.origin;
      if (first + j * HP_PROBE_THREADS >= end || distance > t.range) continue;   // (outside the range: cannot happen, origin and range come from the column's extent)
      const uint32_t slot = distance >> t.radix_bits;
      atomicOr(&s_bits[slot >> 5], 1u << (slot & 31));
      if (t.bloom_bits) {
        const uint32_t bit = (key & (BLOOM_BITS - 1)) >> t.radix_bits;
        if (bloom_in_lds) atomicOr(&s_bloom[bit >> 5], 1u << (bit & 31));
        else atomicOr(t.bloom_bits + static_cast<size_t>(partition) * bloom_words + (bit >> 5), 1u << (bit & 31));
      }
    }
  }
  __syncthreads();
  uint32_t* out = t.partial_bits + static_cast<size_t>(block) * t.words;
  for (uint32_t i = tid; i < t.words; i += HP_PROBE_THREADS) out[i] = s_bits[i];
  if (bloom_in_lds) {
    uint32_t* bloom_out = t.partial_bloom + static_cast<size_t>(block) * bloom_words;
    for (uint32_t i = tid; i < bloom_words; i += HP_PROBE_THREADS) bloom_out[i] = s_bloom[i];
  }
}

__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_ranks(HpLayout layout, HpTable t) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hp_smem[];   // [words] the partition's presence bits
  __shared__ uint32_t s_tmp[16];
  const uint32_t tid = threadIdx.x, partition = blockIdx.x;
  u32x2_t* entries = t.entries + static_cast<size_t>(partition) * t.words;
  const uint32_t first_share = layout.mark_first[partition], end_share = layout.mark_first[partition + 1];
  for (uint32_t w = tid; w < t.words; w += HP_PROBE_THREADS) {   // (neighbouring threads, neighbouring words: every share's table is read in full lines)
    uint32_t bits = 0;
    for (uint32_t share = first_share; share < end_share; ++share) bits |= t.partial_bits[static_cast<size_t>(share) * t.words + w];
    hp_smem[w] = bits;
  }
  __syncthreads();
  const uint32_t per = (t.words + HP_PROBE_THREADS - 1) / HP_PROBE_THREADS;   // thread i: words [i * per, (i + 1) * per)
  uint32_t mine = 0;
  for (uint32_t j = 0; j < per; ++j) { const uint32_t w = tid * per + j; mine += w < t.words ? __popc(hp_smem[w]) : 0u; }
  uint32_t total = 0;
  uint32_t running = hp_block_exclusive_scan(mine, s_tmp, tid, &total);
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t w = tid * per + j;
    if (w < t.words) { const uint32_t bits = hp_smem[w]; entries[w] = u32x2_t{bits, running}; running += __popc(bits); }
  }
  // fewer distinct keys than tuples: a key twice (fine for an existence-only table, the end of this path otherwise)
  if (tid == 0 && !t.existence_only && total != layout.build_off[partition + 1] - layout.build_off[partition]) atomicOr(t.flags, 1u);
  const uint32_t bloom_words = hp_bloom_words(t.radix_bits);
  if (t.bloom_bits && bloom_words <= HP_BLOOM_LDS_WORDS) {
    for (uint32_t w = tid; w < bloom_words; w += HP_PROBE_THREADS) {
      uint32_t bits = 0;
      for (uint32_t share = first_share; share < end_share; ++share) bits |= t.partial_bloom[static_cast<size_t>(share) * bloom_words + w];
      t.bloom_bits[static_cast<size_t>(partition) * bloom_words + w] = bits;
    }
  }
}

__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_ids(HpSide build, HpLayout layout, HpTable t) {
  const uint32_t tid = threadIdx.x, partitions = 1u << t.radix_bits;
  uint32_t partition, begin, end;
  if (!hp_mark_share(layout, partitions, hp_logical_block(), &partition, &begin, &end)) return;
  const u32x2_t* entries = t.entries + static_cast<size_t>(partition) * t.words;
  uint32_t* ids = t.ids + layout.build_off[partition];
  // four tuples per thread and round, their loads in flight together (tuple -> entry -> store is two dependent round trips)
  for (uint32_t first = begin + tid; first < end; first += 4 * HP_PROBE_THREADS) {
    u32x2_t tuple[4], entry[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) { const uint32_t i = first + j * HP_PROBE_THREADS; tuple[j] = i < end ? build.tuples[i] : u32x2_t{t.origin, 0u}; }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) { const uint32_t distance = tuple[j].x - t.origin; entry[j] = entries[distance <= t.range ? (distance >> t.radix_bits) >> 5 : 0u]; }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t distance = tuple[j].x - t.origin, slot = distance >> t.radix_bits;
      if (first + j * HP_PROBE_THREADS < end && distance <= t.range) ids[entry[j].y + __popc(entry[j].x & ((1u << (slot & 31)) - 1))] = tuple[j].y;
    }
  }
}

// ---- the probe: tables staged in LDS, per-wave probing ---------------------------------------------------------------------------
// The group (partition, or probe chunk without radix partitioning) that holds probe step `step`: the last one whose first step is <= step.
__device__ __forceinline__ uint32_t hp_group_of_step(const HpLayout& layout, uint32_t step) {
  uint32_t lo = 0, hi = layout.n_groups;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (layout.first_step[mid] <= step) lo = mid; else hi = mid; }
  return lo;
}

__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_probe(HpProbe a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hp_smem[];
  __shared__ uint32_t s_count;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const HpTable& t = a.table;
  u32x2_t* s_entries = reinterpret_cast<u32x2_t*>(hp_smem);   // [words]
  // Every workgroup takes the same number of consecutive steps, wherever the partitions' borders fall (dbgen's order keys populate a quarter
  // of the partitions: a fixed number of workgroups per partition would leave three quarters of the device idle); it stages a partition's
  // table when its steps enter the partition -- once, or a few times.
  const uint32_t n_steps = a.layout.first_step[a.layout.n_groups];
  const uint32_t per = (n_steps + gridDim.x - 1) / gridDim.x;
  const uint32_t first_step = hp_logical_block() * per, end_step = first_step + per < n_steps ? first_step + per : n_steps;   // (a partition's steps: one XCD, one L2)
  if (tid == 0) s_count = 0;
  uint32_t staged = 0xFFFFFFFFu;
  const uint32_t bloom_words = hp_bloom_words(t.radix_bits);
  for (uint32_t step = first_step; step < end_step; ++step) {
    const u32x4_t info = a.layout.steps[2 * step];
    const uint32_t partition = info.w;
    if (partition != staged) {
      __syncthreads();   // (the table of the partition before is no longer read)
      const u32x2_t* entries = t.entries + static_cast<size_t>(partition) * t.words;
      for (uint32_t i = tid; i < t.words; i += HP_PROBE_THREADS) s_entries[i] = entries[i];
      staged = partition;
      __syncthreads();
    }
    const uint32_t step_begin = info.x, probe_end = info.y;
    const uint32_t* bloom = a.bloom_bits ? a.bloom_bits + static_cast<size_t>(partition) * bloom_words : nullptr;   // the partition's slice of the filter
    uint32_t counted = 0;   // materialised elements | pairs << 16 of this thread's tuples
    uint32_t word[HP_PER_THREAD];
#pragma unroll
    for (uint32_t q = 0; q < HP_PER_THREAD; ++q) {
      const uint32_t g = step_begin + tid * HP_PER_THREAD + q;   // (a thread's tuples are neighbours: hp_emit reads them the same way)
      word[q] = 0;
      if (g >= probe_end) continue;
      const uint32_t key = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.tuples + g));
      const uint32_t distance = key - t.origin;
      const uint32_t slot = distance >> t.radix_bits;
      const u32x2_t entry = s_entries[distance <= t.range ? slot >> 5 : 0u];
      const bool found = distance <= t.range && ((entry.x >> (slot & 31)) & 1);
      bool materialised = true;
      if (!found && bloom && !a.keep_nulls) { const uint32_t bit = (key & (BLOOM_BITS - 1)) >> t.radix_bits; materialised = (bloom[bit >> 5] >> (bit & 31)) & 1; }   // join_hash_steps.hpp:354-358
      bool null_partner;
      const bool emits = materialised && pk_emits<false>(a.mode, found, &null_partner);
      const uint32_t rank = entry.y + __popc(entry.x & ((1u << (slot & 31)) - 1));
      word[q] = (materialised ? HP_MATERIALISED : 0u) | (emits ? HP_EMITS : 0u) | (found ? HP_FOUND | (rank & HP_RANK) : 0u);
      counted += (materialised ? 1u : 0u) + (emits ? 0x10000u : 0u);
    }
    const uint32_t g0 = step_begin + tid * HP_PER_THREAD;
    if (g0 + HP_PER_THREAD <= probe_end && (g0 & 3) == 0) {
      *reinterpret_cast<u32x4_t*>(a.results + g0) = u32x4_t{word[0], word[1], word[2], word[3]};
    } else {
#pragma unroll
      for (uint32_t q = 0; q < HP_PER_THREAD; ++q) if (g0 + q < probe_end) a.results[g0 + q] = word[q];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) counted += __shfl_xor(counted, d, 64);
    if (lane == 0) atomicAdd(&s_count, counted);
    __syncthreads();
    if (tid == 0) { a.step_counts[step] = s_count; s_count = 0; }
    __syncthreads();
  }
}

// ---- the plan of the output (one workgroup) ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_plan(HpProbe a) {
  __shared__ uint32_t s_tmp[16];
  const uint32_t tid = threadIdx.x;
  const uint32_t n_groups = a.layout.n_groups;
  const uint32_t n_steps = a.layout.first_step[n_groups];
  // running sums over the steps; element_base first holds the elements of ALL earlier steps
  uint32_t pairs_before = 0, elements_before = 0;
  for (uint32_t begin = 0; begin < n_steps; begin += HP_PROBE_THREADS) {
    const uint32_t step = begin + tid;
    const uint32_t packed = step < n_steps ? a.step_counts[step] : 0u;
    uint32_t total_pairs = 0, total_elements = 0;
    const uint32_t pairs = hp_block_exclusive_scan(packed >> 16, s_tmp, tid, &total_pairs);
    const uint32_t elements = hp_block_exclusive_scan(packed & 0xFFFFu, s_tmp, tid, &total_elements);
    if (step < n_steps) { a.pair_base[step] = pairs_before + pairs; a.element_base[step] = elements_before + elements; }
    pairs_before += total_pairs;
    elements_before += total_elements;
  }
  __syncthreads();   // (this workgroup's own global stores above are read below)
  // PosLists per group: a new one every 131 070 materialised elements (join_hash_steps.hpp:47,541-591)
  uint32_t slices_before = 0;
  for (uint32_t begin = 0; begin < n_groups; begin += HP_PROBE_THREADS) {
    const uint32_t g = begin + tid;
    uint32_t slices = 0;
    if (g < n_groups) {
      const uint32_t first = a.layout.first_step[g], next = a.layout.first_step[g + 1];
      const uint32_t elements_begin = first < n_steps ? a.element_base[first] : elements_before, elements_end = next < n_steps ? a.element_base[next] : elements_before;
      a.group_elements[g] = elements_begin;
      slices = (elements_end - elements_begin + PROBE_SIZE_PER_CHUNK - 1) / PROBE_SIZE_PER_CHUNK;
    }
    uint32_t total = 0;
    const uint32_t before = slices_before + hp_block_exclusive_scan(slices, s_tmp, tid, &total);
    if (g < n_groups) a.slice_base[g] = before;
    slices_before += total;
  }
  const uint32_t n_slices = slices_before;
  __syncthreads();
  // elements inside the group: take the running sum at the group's first step off
  for (uint32_t begin = 0; begin < n_steps; begin += HP_PROBE_THREADS) {
    const uint32_t step = begin + tid;
    if (step < n_steps) {
      const u32x4_t info = a.layout.steps[2 * step];
      a.element_base[step] -= a.group_elements[info.z];
      a.layout.steps[2 * step + 1] = u32x4_t{a.pair_base[step], a.element_base[step], a.slice_base[info.z], a.layout.build_off[info.w]};   // (slice_base: this workgroup's own stores, behind the barrier above)
    }
  }
  if (tid == 0) {
    a.slice_base[n_groups] = n_slices;
    const uint64_t n_pairs = pairs_before;
    const uint32_t twice = *a.table.flags & 1u;   // (hp_table: a build key twice -- nothing may be written, the host runs the general kernels)
    const uint32_t fits = !twice && n_pairs <= a.capacity && n_slices <= a.slice_capacity ? 1u : 0u;
    a.plan->fits = fits;
    a.plan->n_slices = n_slices;
    if (fits && a.slice_offsets) a.slice_offsets[n_slices] = n_pairs;
    if (a.status) {
      a.status->n_pairs = n_pairs;
      a.status->n_slices = n_slices;
      a.status->fits = fits;
      a.status->build_confirmed = twice ? 0u : 1u;
      a.status->error = 0;
      a.status->reserved = 0;
    }
    a.mailbox->n_pairs = n_pairs;
    a.mailbox->n_slices = n_slices;
    a.mailbox->n_uncached = 0;
    a.mailbox->fits = fits;
    a.mailbox->build_unconfirmed = 0;
    a.mailbox->duplicate = twice;
    __threadfence_system();
  }
}

// ---- the pairs ----------------------------------------------------------------------------------------------------------------------
// What hp_emit knows about a step before it works on it: requested a step ahead.
struct HpEmitLoads {
  u32x4_t info, plan;
  uint32_t word[HP_PER_THREAD], row[HP_PER_THREAD];
};
__device__ __forceinline__ void hp_emit_load(const HpProbe& a, uint32_t step, uint32_t tid, HpEmitLoads& l) {
  l.info = a.layout.steps[2 * step];
  l.plan = a.layout.steps[2 * step + 1];
  const uint32_t g0 = l.info.x + tid * HP_PER_THREAD;
  if (g0 + HP_PER_THREAD <= l.info.y && (g0 & 3) == 0) {   // (the common case: one 16-byte and two 16-byte loads)
    // (read once: nontemporal, so that the partition's rank -> RowID slice stays in the L2 while 1.7 GB stream through it)
    const u32x4_t words = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(a.results + g0));
    const u32x4_t t0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(a.tuples + g0)), t1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(a.tuples + g0 + 2));
    l.word[0] = words.x; l.word[1] = words.y; l.word[2] = words.z; l.word[3] = words.w;
    l.row[0] = t0.y; l.row[1] = t0.w; l.row[2] = t1.y; l.row[3] = t1.w;
  } else {
#pragma unroll
    for (uint32_t q = 0; q < HP_PER_THREAD; ++q) {
      l.word[q] = g0 + q < l.info.y ? a.results[g0 + q] : 0u;
      l.row[q] = g0 + q < l.info.y ? a.tuples[g0 + q].y : 0u;
    }
  }
}

// Persistent workgroups, each with a run of consecutive steps (a partition's steps in one L2, see hp_logical_block); a step's result words and
// RowIDs are requested while the step before is written.
__global__ __launch_bounds__(HP_PROBE_THREADS) void hp_emit(HpProbe a) {
  __shared__ uint32_t s_tmp[16];
  const uint32_t tid = threadIdx.x;
  if (!a.plan->fits) return;
  const uint32_t n_steps = a.layout.first_step[a.layout.n_groups];
  const uint32_t per = (n_steps + gridDim.x - 1) / gridDim.x;
  const uint32_t first_step = hp_logical_block() * per, end_step = first_step + per < n_steps ? first_step + per : n_steps;
  if (first_step >= end_step) return;
  u32x2_t* probe_out = reinterpret_cast<u32x2_t*>(a.probe_out);
  u32x2_t* build_out = reinterpret_cast<u32x2_t*>(a.build_out);
  HpEmitLoads current, ahead;
  hp_emit_load(a, first_step, tid, current);
  for (uint32_t step = first_step; step < end_step; ++step) {
    if (step + 1 < end_step) hp_emit_load(a, step + 1, tid, ahead);
    // a thread's tuples are neighbours: ONE prefix over the threads (materialised elements and pairs in one word), the rest inside the thread
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < HP_PER_THREAD; ++q) mine += ((current.word[q] & HP_MATERIALISED) ? 1u : 0u) + ((current.word[q] & HP_EMITS) ? 0x10000u : 0u);
    uint32_t unused_total = 0;
    const uint32_t before = hp_block_exclusive_scan(mine, s_tmp, tid, &unused_total);
    uint32_t pairs_at = current.plan.x + (before >> 16), elements_at = current.plan.y + (before & 0xFFFFu);
    const uint32_t build_begin = current.plan.w;
    uint32_t partner_id[HP_PER_THREAD];
    if (build_out) {
#pragma unroll
      for (uint32_t q = 0; q < HP_PER_THREAD; ++q) partner_id[q] = (current.word[q] & HP_FOUND) ? a.table.ids[build_begin + (current.word[q] & HP_RANK)] : 0xFFFFFFFFu;
    }
    const bool all_four = (current.word[0] & current.word[1] & current.word[2] & current.word[3] & HP_EMITS) != 0 && (pairs_at & 1) == 0;
#pragma unroll
    for (uint32_t q = 0; q < HP_PER_THREAD; ++q) {   // a new PosList every 131 070 materialised elements of the group: it begins with this tuple's pairs
      if (!(current.word[q] & HP_MATERIALISED)) continue;
      uint32_t pairs_before_tuple = pairs_at;
#pragma unroll
      for (uint32_t e = 0; e < q; ++e) pairs_before_tuple += (current.word[e] & HP_EMITS) ? 1u : 0u;
      if (elements_at % PROBE_SIZE_PER_CHUNK == 0) a.slice_offsets[current.plan.z + elements_at / PROBE_SIZE_PER_CHUNK] = pairs_before_tuple;
      ++elements_at;
    }
    if (all_four) {   // (the common case: the thread's four pairs as two aligned 16-byte nontemporal stores per PosList)
#pragma unroll
      for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t r0 = current.row[2 * h], r1 = current.row[2 * h + 1];
        __builtin_nontemporal_store(u32x4_t{r0 >> 16, r0 & 0xFFFFu, r1 >> 16, r1 & 0xFFFFu}, reinterpret_cast<u32x4_t*>(probe_out + pairs_at + 2 * h));
        if (build_out) {
          const uint32_t w0 = current.word[2 * h], w1 = current.word[2 * h + 1], p0 = partner_id[2 * h], p1 = partner_id[2 * h + 1];
          __builtin_nontemporal_store(u32x4_t{(w0 & HP_FOUND) ? p0 >> 16 : 0xFFFFFFFFu, (w0 & HP_FOUND) ? p0 & 0xFFFFu : 0xFFFFFFFFu, (w1 & HP_FOUND) ? p1 >> 16 : 0xFFFFFFFFu,
                                               (w1 & HP_FOUND) ? p1 & 0xFFFFu : 0xFFFFFFFFu}, reinterpret_cast<u32x4_t*>(build_out + pairs_at + 2 * h));
        }
      }
    } else {
#pragma unroll
      for (uint32_t q = 0; q < HP_PER_THREAD; ++q) {
        const uint32_t word = current.word[q];
        if (!(word & HP_EMITS)) continue;
        probe_out[pairs_at] = u32x2_t{current.row[q] >> 16, current.row[q] & 0xFFFFu};
        if (build_out) build_out[pairs_at] = (word & HP_FOUND) ? u32x2_t{partner_id[q] >> 16, partner_id[q] & 0xFFFFu} : u32x2_t{0xFFFFFFFFu, 0xFFFFFFFFu};
        ++pairs_at;
      }
    }
    current = ahead;
  }
}
