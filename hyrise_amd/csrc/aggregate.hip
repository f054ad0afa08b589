// aggregate.hip -- AggregateHash on MI355X: GROUP BY over up to 4 integer (or dictionary-named) columns with
// MIN / MAX / SUM / AVG / COUNT / COUNT(*) / COUNT(DISTINCT) / STDDEV_SAMP / ANY aggregates.
//
// What it replaces (reference, CPU):
//   AggregateHash::_on_execute / _aggregate             operators/aggregate_hash.cpp:950-1372
//   _partition_by_groupby_keys (key derivation)          aggregate_hash.cpp:661-948
//   get_or_add_result + _aggregate_segment (hot loop)    aggregate_hash.cpp:317-403, 605-655
//   WindowFunctionBuilder accumulators                   operators/abstract_aggregate_operator.hpp:30-133
//
// The reference walks the table row by row on one thread; its result order is "first occurrence of the group"
// (or ascending key under the immediate-key shortcut).  Device design:
//   aggregate_rows   one workgroup per 8192-row slice.  Every row's GROUP BY tuple (NULL mask + raw 64-bit values) is
//                    looked up in a workgroup-private open-addressed hash table staged in LDS (tag word = lock,
//                    ds_cmpswap to claim a slot).  Rows of the slice's first four groups accumulate in registers,
//                    aggregate by aggregate, and reach the table through a wave reduction + one LDS atomic per wave;
//                    rows of further groups use LDS atomics (ds_add_u64 / ds_add_f64 / ds_min / ds_max).  The smallest
//                    and largest global row number of every group ride along.  At the end the (few) occupied LDS
//                    slots are merged into a global open-addressed table with agent-scope atomics -- for TPC-H Q1 that
//                    is 4 groups x 6 aggregates per workgroup instead of 8192 x 6 global atomics.  Rows whose group
//                    does not fit the LDS table go to the global table directly.
//   compact_groups   occupied global slots -> dense arrays.
// The host then orders the groups exactly like the reference (first-row rank, or key order) and derives AVG.
// SUM/AVG over float/double columns use f64 atomics: the additions happen in a different order than the reference's
// sequential loop, so those results are compared with a stated tolerance (1e-9 relative); everything else is exact.
#include "hy_device.hpp"
#include "hy_decode.hpp"
#include "hy_arithmetic.hpp"
#include "hy_scan_job.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <thread>
#include <type_traits>
#include <vector>

// Two builds of this file: GROUP BY over at most 4 columns (this translation unit: a tuple is five 64-bit words in registers -- every
// kernel below is tuned at that size) and, for the plans with more (TPC-H Q10 groups by seven columns, Q18 by five; the reference takes any
// number: aggregate_hash.cpp:1184-1198, AggregateKeySmallVector), the same code with tuples of nine words (aggregate_wide.hip includes this
// file with HY_MAX_GROUPBY 8; its entry points carry the suffix _wide and the ones below hand wider GROUP BYs over to them), and for nine to
// sixteen columns a third time with seventeen words (aggregate_widest.hip, suffix _widest).
#ifndef HY_MAX_GROUPBY
#define HY_MAX_GROUPBY 4
#endif
#if HY_MAX_GROUPBY == 4
#define HY_AGG_NAMESPACE narrow_keys
#define HY_AGG_ENTRY(name) name
#define HY_AGG_NEXT(name) name##_wide       // the build that takes what this one does not
#elif HY_MAX_GROUPBY == 8
#define HY_AGG_NAMESPACE wide_keys
#define HY_AGG_ENTRY(name) name##_wide
#define HY_AGG_NEXT(name) name##_widest     // (aggregate_widest.hip: nine to sixteen GROUP BY columns, seventeen-word tuples)
#else
#define HY_AGG_NAMESPACE widest_keys
#define HY_AGG_ENTRY(name) name##_widest
#endif

namespace hy {
#if HY_MAX_GROUPBY == 4
thread_local uint32_t t_aggregate_recommended = 0, t_aggregate_next_path = 0;   // hy_device.hpp
#endif
inline namespace HY_AGG_NAMESPACE {

constexpr uint32_t MAX_GROUPBY = HY_MAX_GROUPBY;
constexpr uint32_t MAX_AGGREGATES = 8;
constexpr uint32_t LDS_SLOTS = 256;
constexpr uint32_t MAX_LDS_PROBES = 24;      // linear probing in a workgroup's table: rows that would walk further are handled like rows of a full table
constexpr uint32_t MAX_GLOBAL_PROBES = 512;   // linear probing in the global group table: longer sequences count as overflow
constexpr uint32_t DENSE_GROUPS = 4;   // slices with at most this many groups accumulate in thread-private LDS cells
constexpr uint32_t TAG_EMPTY = 0, TAG_LOCKED = 1;   // ready tags have bit 31 set
constexpr uint32_t AGG_SUM_SQUARES = 100;           // device-internal: sum of (x - pivot)^2 as double (second accumulator of STDDEV_SAMP)
constexpr uint32_t AGG_SUM_SHIFTED = 101;           // device-internal: sum of (x - pivot) as double + count (first accumulator of STDDEV_SAMP)

struct AggColumn {
  const DevSegment* segments;
  uint32_t function;     // HY_AGG_* (aggregates only)
  uint32_t data_type;    // HY_TYPE_*
  uint32_t is_float;     // accumulate as double
  uint32_t reserved;
  double pivot;          // AGG_SUM_SHIFTED / AGG_SUM_SQUARES: a value of the column, subtracted before accumulating (see STDDEV_SAMP below)
};

struct AggArgs {
  AggColumn groupby[MAX_GROUPBY];
  AggColumn aggregates[MAX_AGGREGATES];
  uint32_t n_groupby;
  uint32_t n_aggregates;
  const Slice* slices;
  const uint64_t* row_base;   // global row number of every chunk's first row
  // global table
  uint32_t capacity;          // power of two
  uint32_t* tags;
  uint64_t* keys;             // [capacity][n_groupby + 1]
  uint64_t* first_row;        // [capacity]
  uint64_t* last_row;         // [capacity]
  uint64_t* values;           // [capacity][n_aggregates]
  uint64_t* counts;           // [capacity][n_aggregates]
  uint32_t* overflow;         // flags: [0] the global table overflowed, [1] number of groups (compact_groups), [2] give up: too many rows left the
                              //        LDS tables (the host switches to the partitioned path), [3] rows that left the LDS tables so far
  uint32_t spill_limit;       // [3] above this sets [2]
  uint64_t* trace;            // debug (HY_AGG_TRACE): 12 wall-clock stamps per slice, else nullptr
  uint32_t pos_cache;         // aggregate_rows: some column is a reference column -- the launch carries LDS for the slice's PosList offsets
};
enum : uint32_t { FLAG_OVERFLOW = 0, FLAG_GROUPS = 1, FLAG_GIVE_UP = 2, FLAG_SPILLED = 3, /* 4, 5: FLAG_PASSED */
                  FLAG_SMALL_REFUSED = 6 /* aggregate_small_domain: a 4-bit counter overflowed, run aggregate_rows */,
                  FLAG_CROWDED = 7 /* aggregate_rows: slices most of whose rows found no place in the LDS table */ };
constexpr uint32_t CROWDED_SLICES = 8;   // ... this many of them end the attempt at once

// order-preserving map double -> int64 (so MIN/MAX of floating point values can use integer atomics)
__device__ __forceinline__ int64_t ordered_bits(double d) {
  int64_t b = __double_as_longlong(d);
  return b < 0 ? b ^ 0x7FFFFFFFFFFFFFFFll : b;
}

// Hash of a GROUP BY tuple: two independent 32-bit multiplicative mixes (64-bit multiplies cost ~10 instructions each
// on this hardware and the hash is computed per row).  Only the placement in the hash tables depends on it.
__device__ __forceinline__ uint64_t hash_tuple(const uint64_t* tuple, uint32_t words) {
  uint32_t h1 = 0x9E3779B9u, h2 = 0x85EBCA6Bu;
  for (uint32_t w = 0; w < words; ++w) {
    const uint32_t lo = static_cast<uint32_t>(tuple[w]), hi = static_cast<uint32_t>(tuple[w] >> 32);
    h1 = (h1 ^ lo) * 0x9E3779B1u;
    h1 = (h1 ^ (h1 >> 15) ^ hi) * 0x85EBCA77u;
    h2 = (h2 ^ hi) * 0xC2B2AE3Du;
    h2 = (h2 ^ (h2 >> 13) ^ lo) * 0x27D4EB2Fu;
  }
  h1 ^= h1 >> 16;
  h2 ^= h2 >> 15;
  return (static_cast<uint64_t>(h2) << 32) | h1;
}

// hash_tuple of a tuple held in registers (static indices only)
__device__ __forceinline__ uint64_t hash_tuple_in_registers(const uint64_t (&tuple)[MAX_GROUPBY + 1], uint32_t words) {
  uint32_t h1 = 0x9E3779B9u, h2 = 0x85EBCA6Bu;
#pragma unroll
  for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) {
    if (w >= words) continue;
    const uint32_t lo = static_cast<uint32_t>(tuple[w]), hi = static_cast<uint32_t>(tuple[w] >> 32);
    h1 = (h1 ^ lo) * 0x9E3779B1u;
    h1 = (h1 ^ (h1 >> 15) ^ hi) * 0x85EBCA77u;
    h2 = (h2 ^ hi) * 0xC2B2AE3Du;
    h2 = (h2 ^ (h2 >> 13) ^ lo) * 0x27D4EB2Fu;
  }
  h1 ^= h1 >> 16;
  h2 ^= h2 >> 15;
  return (static_cast<uint64_t>(h2) << 32) | h1;
}

// Hash for the workgroup's LDS table only (computed for every row): fold the words with rotates and xors, then two
// multiplies -- 32-bit multiplies run at quarter rate, the per-word mixing of hash_tuple costs more than the lookup itself.
// Structured tuples may collide here; that costs probes in a 256-slot table, never correctness.
__device__ __forceinline__ uint64_t hash_local(const uint64_t (&tuple)[MAX_GROUPBY + 1], uint32_t words) {
  uint32_t lo = 0x9E3779B9u, hi = 0x85EBCA6Bu;
#pragma unroll
  for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) {
    if (w < words) {
      lo = ((lo << 7) | (lo >> 25)) ^ static_cast<uint32_t>(tuple[w]);
      hi = ((hi << 11) | (hi >> 21)) + static_cast<uint32_t>(tuple[w] >> 32);
    }
  }
  uint32_t h1 = (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA77u;
  h1 ^= h1 >> 15;
  const uint32_t h2 = (h1 ^ lo) * 0xC2B2AE3Du;
  return (static_cast<uint64_t>(h2 ^ (h2 >> 13)) << 32) | h1;
}

__device__ __forceinline__ uint64_t initial_value(uint32_t function) {
  if (function == HY_AGG_MIN) return static_cast<uint64_t>(INT64_MAX);
  if (function == HY_AGG_MAX) return static_cast<uint64_t>(INT64_MIN);
  return 0;
}

// Find or insert `tuple` in the GLOBAL table; returns the slot or 0xFFFFFFFF when the table is full.
// Lock discipline (also for the LDS table below): a lane NEVER spins in an inner loop on a slot another lane may hold --
// lanes of one wave run in lockstep, so the holder could be masked off behind the spinning lane forever.  Every
// iteration of the single retry loop either finishes the whole critical section (claim, initialise, publish) or makes
// no blocking step at all; a lane that finds a slot locked simply goes around again.
__device__ __forceinline__ uint32_t global_slot(const AggArgs& a, const uint64_t* tuple, uint32_t words, uint64_t hash) {
  const uint32_t ready = 0x80000000u | static_cast<uint32_t>(hash >> 33);
  uint32_t slot = static_cast<uint32_t>(hash) & (a.capacity - 1);
  uint32_t probes = 0;
  uint32_t result = 0xFFFFFFFFu;
  bool done = false;
  while (!done) {
    // Relaxed, like the key words below: agent-scope atomic loads are served by the L2, which already holds the key
    // words when it shows a published tag (the publisher's release store waits for them), and a wave's loads return in
    // order; the signal fence keeps the compiler from moving the key loads up.  An acquire load here costs an L1
    // invalidation per probe -- per ROW on the many-groups path.
    uint32_t tag = __hip_atomic_load(&a.tags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_ACQUIRE);
    if (tag == TAG_EMPTY) {
      uint32_t expected = TAG_EMPTY;
      if (__hip_atomic_compare_exchange_strong(&a.tags[slot], &expected, TAG_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        for (uint32_t w = 0; w < words; ++w) __hip_atomic_store(&a.keys[static_cast<size_t>(slot) * words + w], tuple[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.first_row[slot], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.last_row[slot], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t g = 0; g < a.n_aggregates; ++g) {
          __hip_atomic_store(&a.values[static_cast<size_t>(slot) * a.n_aggregates + g], initial_value(a.aggregates[g].function), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&a.counts[static_cast<size_t>(slot) * a.n_aggregates + g], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&a.tags[slot], ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        result = slot;
        done = true;
      }
      // lost the race: look at the same slot again next time round
    } else if (tag != TAG_LOCKED) {
      bool equal = tag == ready;
      if (equal) {
        for (uint32_t w = 0; w < words; ++w) equal &= __hip_atomic_load(&a.keys[static_cast<size_t>(slot) * words + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tuple[w];
      }
      if (equal) {
        result = slot;
        done = true;
      } else {
        slot = (slot + 1) & (a.capacity - 1);
        // a probe sequence this long means a table that is (nearly) full: report overflow, the host retries with a larger
        // one -- walking a 95 % full table of millions of slots group by group takes minutes
        if (++probes >= (a.capacity < MAX_GLOBAL_PROBES ? a.capacity : MAX_GLOBAL_PROBES)) done = true;
      }
    } else {
      __builtin_amdgcn_s_sleep(1);   // locked by someone else: retry
    }
  }
  return result;
}

__device__ __forceinline__ void merge_global(const AggArgs& a, uint32_t slot, uint32_t g, uint64_t value, uint64_t count) {
  if (count == 0) return;
  const AggColumn& c = a.aggregates[g];
  uint64_t* target = &a.values[static_cast<size_t>(slot) * a.n_aggregates + g];
  switch (c.function) {
    case HY_AGG_MIN: atomicMin(reinterpret_cast<long long*>(target), static_cast<long long>(value)); break;
    case HY_AGG_MAX: atomicMax(reinterpret_cast<long long*>(target), static_cast<long long>(value)); break;
    case HY_AGG_SUM:
    case HY_AGG_AVG:
    case AGG_SUM_SHIFTED:
    case AGG_SUM_SQUARES:
      if (c.is_float || c.function != HY_AGG_SUM) atomicAdd(reinterpret_cast<double*>(target), __longlong_as_double(static_cast<long long>(value)));
      else atomicAdd(reinterpret_cast<unsigned long long*>(target), static_cast<unsigned long long>(value));
      break;
    default: break;
  }
  atomicAdd(reinterpret_cast<unsigned long long*>(&a.counts[static_cast<size_t>(slot) * a.n_aggregates + g]), static_cast<unsigned long long>(count));
}

// accumulator (+)= contribution, as plain arithmetic on 64-bit words (the atomics below do the same on shared cells)
__device__ __forceinline__ uint64_t combine(const AggColumn& c, uint64_t accumulator, uint64_t contribution) {
  switch (c.function) {
    case HY_AGG_MIN: return static_cast<uint64_t>(min(static_cast<long long>(accumulator), static_cast<long long>(contribution)));
    case HY_AGG_MAX: return static_cast<uint64_t>(max(static_cast<long long>(accumulator), static_cast<long long>(contribution)));
    case HY_AGG_SUM:
    case HY_AGG_AVG:
    case AGG_SUM_SHIFTED:
    case AGG_SUM_SQUARES:
      if (c.is_float || c.function != HY_AGG_SUM) {
        return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(accumulator)) + __longlong_as_double(static_cast<long long>(contribution))));
      }
      return accumulator + contribution;
    default: return accumulator;
  }
}

// decoded value -> the word the accumulators of this aggregate work on
__device__ __forceinline__ uint64_t contribution_from(const AggColumn& c, uint64_t bits) {
  switch (c.function) {
    case HY_AGG_MIN:
    case HY_AGG_MAX: return c.is_float ? static_cast<uint64_t>(ordered_bits(__longlong_as_double(static_cast<long long>(bits)))) : bits;
    case HY_AGG_SUM: return bits;
    case HY_AGG_AVG: return c.is_float ? bits : static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<int64_t>(bits))));
    case AGG_SUM_SHIFTED:
    case AGG_SUM_SQUARES: {
      const double x = (c.is_float ? __longlong_as_double(static_cast<long long>(bits)) : static_cast<double>(static_cast<int64_t>(bits))) - c.pivot;
      return static_cast<uint64_t>(__double_as_longlong(c.function == AGG_SUM_SQUARES ? x * x : x));
    }
    default: return 0;
  }
}

__device__ __forceinline__ void accumulate_lds(const AggColumn& c, uint64_t* target, uint64_t contribution) {
  switch (c.function) {
    case HY_AGG_MIN: atomicMin(reinterpret_cast<long long*>(target), static_cast<long long>(contribution)); break;
    case HY_AGG_MAX: atomicMax(reinterpret_cast<long long*>(target), static_cast<long long>(contribution)); break;
    case HY_AGG_SUM:
    case HY_AGG_AVG:
    case AGG_SUM_SHIFTED:
    case AGG_SUM_SQUARES:
      if (c.is_float || c.function != HY_AGG_SUM) atomicAdd(reinterpret_cast<double*>(target), __longlong_as_double(static_cast<long long>(contribution)));
      else atomicAdd(reinterpret_cast<unsigned long long*>(target), static_cast<unsigned long long>(contribution));
      break;
    default: break;
  }
}

// Reductions over the 64 lanes of a wave with DPP moves (a few cycles each; __shfl_xor is a ds_bpermute round trip through
// the LDS crossbar per step -- the eleven reductions per accumulator used to cost more than the accumulation itself).
// Scan-shaped: row_shr 1, 2, 4, 8 inside every row of 16 lanes, then row_bcast 15 / 31; the result is in LANE 63.  Lanes
// without a source (and rows a broadcast does not reach) combine with `identity`.
template <typename Combine>
__device__ __forceinline__ uint64_t wave_reduce_to_lane63(uint64_t v, uint64_t identity, Combine combine_values) {
  const int id_lo = static_cast<int>(static_cast<uint32_t>(identity)), id_hi = static_cast<int>(static_cast<uint32_t>(identity >> 32));
#define HY_DPP_STEP(CTRL, ROW_MASK)                                                                                         \
  {                                                                                                                        \
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id_lo, static_cast<int>(static_cast<uint32_t>(v)), CTRL, ROW_MASK, 0xF, false));       \
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id_hi, static_cast<int>(static_cast<uint32_t>(v >> 32)), CTRL, ROW_MASK, 0xF, false)); \
    v = combine_values(v, static_cast<uint64_t>(hi) << 32 | lo);                                                            \
  }
  HY_DPP_STEP(0x111, 0xF)   // row_shr:1
  HY_DPP_STEP(0x112, 0xF)   // row_shr:2
  HY_DPP_STEP(0x114, 0xF)   // row_shr:4
  HY_DPP_STEP(0x118, 0xF)   // row_shr:8
  HY_DPP_STEP(0x142, 0xA)   // row_bcast:15 into rows 1 and 3
  HY_DPP_STEP(0x143, 0xC)   // row_bcast:31 into rows 2 and 3
#undef HY_DPP_STEP
  return v;
}

__device__ __forceinline__ uint32_t wave_reduce_u32_to_lane63(uint32_t v, uint32_t identity, bool take_min, bool take_max) {
  return static_cast<uint32_t>(wave_reduce_to_lane63(v, identity, [=](uint64_t a, uint64_t b) {
    const uint32_t x = static_cast<uint32_t>(a), y = static_cast<uint32_t>(b);
    return static_cast<uint64_t>(take_min ? min(x, y) : take_max ? max(x, y) : x + y);
  }));
}

// Aggregate input of one row (generic, one row at a time): false for NULL (NULL inputs leave the aggregate unchanged,
// aggregate_hash.cpp:627-637).
__device__ __forceinline__ bool contribution_of(const AggColumn& c, uint32_t chunk, uint32_t row, uint64_t* contribution) {
  *contribution = 0;
  if (!c.segments) return true;   // COUNT(*)
  const Value v = column_value(c.segments, chunk, row);
  if (v.is_null) return false;
  const uint64_t bits = c.is_float ? static_cast<uint64_t>(__double_as_longlong(c.data_type == HY_TYPE_FLOAT ? static_cast<double>(static_cast<float>(v.f)) : v.f))
                                   : static_cast<uint64_t>(v.i);
  *contribution = contribution_from(c, bits);
  return true;
}

// GROUP BY tuple of one row (generic, one row at a time): word 0 = NULL mask, then the raw values (0 for NULL)
__device__ __forceinline__ void row_tuple(const AggArgs& a, uint32_t chunk, uint32_t row, uint64_t (&tuple)[MAX_GROUPBY + 1]) {
  tuple[0] = 0;
  for (uint32_t g = 0; g < a.n_groupby; ++g) {
    const Value v = column_value(a.groupby[g].segments, chunk, row);
    uint64_t bits = 0;
    if (v.is_null) tuple[0] |= 1ull << g;
    else if (a.groupby[g].is_float) {
      const double d = a.groupby[g].data_type == HY_TYPE_FLOAT ? static_cast<double>(static_cast<float>(v.f)) : v.f;
      bits = d == 0.0 ? 0ull : static_cast<uint64_t>(__double_as_longlong(d));
    } else bits = static_cast<uint64_t>(v.i);
    tuple[g + 1] = bits;
  }
  for (uint32_t g = a.n_groupby; g < MAX_GROUPBY; ++g) tuple[g + 1] = 0;
}

// LDS layout (dynamic): keys[LDS_SLOTS][words] u64 | first[LDS_SLOTS] u64 | last[LDS_SLOTS] u64 | values[LDS_SLOTS][A] u64 |
//                       counts[LDS_SLOTS][A] u32 | tags[LDS_SLOTS] u32 | dense index of a slot [LDS_SLOTS] u32 |
//                       slot of a dense index [DENSE_GROUPS] u32 | number of groups u32 | slot of every row [SLICE_ROWS] u8
//
// One workgroup per 8192-row slice, 32 rows per thread (row k*256 + tid of the slice):
//   pass 1  GROUP BY, four rows of a thread at a time: every tuple is looked up / inserted in the workgroup's LDS table and
//           the row's slot is remembered.  Dictionary columns are keyed by their VALUE IDS inside the slice (one chunk) and
//           translated to values only when the slice's groups are merged into the global table; when every GROUP BY
//           column is a dictionary segment and the product of (dictionary size + 1) fits the table, the combined value-id
//           code IS the slot (no hash, no probing, no key comparison: TPC-H Q1).  A new group also gets a dense index
//           (its arrival number).
//   pass 2  aggregate by aggregate, 16 rows at a time: rows of the first DENSE_GROUPS groups of the slice (TPC-H Q1 has 4
//           in total) accumulate in thread-private LDS cells (one per dense group and thread: LDS atomics that never
//           conflict and whose results nobody waits for); a wave reduction and one LDS atomic per wave and group fold the
//           cells into the table.  (256 threads doing LDS atomics on 4 shared cells serialise completely.)
//   pass 3  only if the slice has more groups: the remaining rows use LDS atomics per row -- the more groups, the fewer
//           conflicts -- and rows whose group does not fit the LDS table go to the global table directly.
//   merge   the slice's groups go to the global table with agent-scope atomics.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) uint32_t global_u32;   // (pointers read from a segment descriptor are generic to the compiler: flat loads)
typedef __attribute__((address_space(1))) u32x2 global_u32x2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 global_u32x4;

// Row k (0..31) of thread `tid` inside its 8192-row slice: four consecutive rows per lane and step, so that the loads of
// rows 4m .. 4m+3 of a value / attribute-vector column are one wide load (16 bytes of floats, a dword of byte value ids).
__device__ __forceinline__ uint32_t slice_row(uint32_t k, uint32_t tid) { return ((k >> 2) * 256 + tid) * 4 + (k & 3); }

// aggregate_rows' PosList cache: the offsets of a thread's four consecutive rows in one LDS word -- the first offset (16 bits) and three
// steps of at most 31 (what a scan's ascending PosList over one chunk gives; anything else -- a NULL RowID, a join's order, a chunk of more
// than 65 536 rows -- is not encodable and the slice reads its PosLists from memory as before).
__device__ __forceinline__ void cached_offsets(uint32_t word, uint32_t (&offset)[4]) {
  offset[0] = word & 0xFFFFu;
  offset[1] = offset[0] + ((word >> 16) & 31u);
  offset[2] = offset[1] + ((word >> 21) & 31u);
  offset[3] = offset[2] + (word >> 26);
}

__global__ __launch_bounds__(256) void aggregate_rows(AggArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t words = a.n_groupby + 1;
  uint64_t* s_keys = reinterpret_cast<uint64_t*>(smem);
  uint64_t* s_first = s_keys + LDS_SLOTS * words;
  uint64_t* s_last = s_first + LDS_SLOTS;
  uint64_t* s_values = s_last + LDS_SLOTS;
  uint32_t* s_counts = reinterpret_cast<uint32_t*>(s_values + LDS_SLOTS * a.n_aggregates);   // [LDS_SLOTS][A]
  uint32_t* s_tags = s_counts + LDS_SLOTS * a.n_aggregates;
  uint32_t* s_dense_of_slot = s_tags + LDS_SLOTS;                                            // [LDS_SLOTS]
  uint32_t* s_slot_of_dense = s_dense_of_slot + LDS_SLOTS;                                   // [DENSE_GROUPS]
  uint32_t* s_n_groups = s_slot_of_dense + DENSE_GROUPS;
  uint32_t* s_present = s_n_groups + 4;                                                      // [8] direct-mapped pass 1: one bit per code met in the slice
  uint32_t* s_spilled = s_present + 8;                                                       // rows of the slice that did not fit the LDS table | the give-up flag as the workgroup saw it
  uint8_t* s_row_slot = reinterpret_cast<uint8_t*>(s_present + 12);                          // [SLICE_ROWS] LDS slot of every row of the slice (a thread's four consecutive rows: one word)
  uint32_t* s_pos = reinterpret_cast<uint32_t*>(s_row_slot + SLICE_ROWS);                      // [SLICE_ROWS / 4] a.pos_cache: the PosList offsets of a thread's four consecutive rows (stage_pos_offsets)
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  for (uint32_t s = tid; s < LDS_SLOTS; s += 256) {
    s_tags[s] = TAG_EMPTY;
    s_first[s] = ~0ull;
    s_last[s] = 0;
    for (uint32_t g = 0; g < a.n_aggregates; ++g) {
      s_values[s * a.n_aggregates + g] = initial_value(a.aggregates[g].function);
      s_counts[s * a.n_aggregates + g] = 0;
    }
  }
  if (tid == 0) {
    *s_n_groups = 0;
    s_spilled[0] = 0;
    s_spilled[1] = __hip_atomic_load(&a.overflow[FLAG_GIVE_UP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_spilled[2] = 0;   // stage_pos_offsets: some block of four rows is not encodable
  }
  __syncthreads();
  if (s_spilled[1]) return;   // the table has too many groups for this kernel: the partitioned path takes over (every thread sees the same word)

  // Workgroup b runs on XCD b % 8, and every XCD has its own L2.  In slice order the eight slices of a chunk would land on
  // eight XCDs and each would fetch the chunk's dictionaries (240 KB for l_extendedprice) for itself; instead, of every 64
  // consecutive slices -- eight chunks of eight slices -- XCD x takes slices 8x .. 8x + 7: one chunk, one L2.
  uint32_t slice_index = blockIdx.x;
  if ((blockIdx.x | 63u) < gridDim.x) slice_index = (blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u);
  const Slice slice = a.slices[slice_index];
  const uint64_t chunk_base = a.row_base[slice.chunk];
  uint64_t* stamps = a.trace ? a.trace + size_t{blockIdx.x} * 12 : nullptr;
  if (stamps && tid == 0) stamps[0] = wall_clock64();
  constexpr uint32_t ROWS = SLICE_ROWS / 256;
  if (slice.row_count == 0) return;
  uint32_t in_table = 0;   // bit k: row k found its group in the LDS table

  // What pass 1 needs of the GROUP BY columns' segment descriptors, once (read through the pointer they are a dependent
  // global load in front of every block's attribute-vector loads: three round trips per block instead of one).
  const void* key_data[MAX_GROUPBY];
  const void* key_dictionary[MAX_GROUPBY];
  uint32_t key_width[MAX_GROUPBY], key_dictionary_size[MAX_GROUPBY], key_type[MAX_GROUPBY];
  const uint32_t* key_pos[MAX_GROUPBY];   // the slice's rows of column g sit behind a single-chunk PosList: value ids are GATHERED at its offsets
  uint32_t local_keys = 0;   // bit g: GROUP BY column g is a dictionary segment in this chunk (keyed by value id inside the slice)
  bool keys_unaligned = false;
#pragma unroll
  for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
    key_data[g] = key_dictionary[g] = nullptr;
    key_pos[g] = nullptr;
    key_width[g] = key_dictionary_size[g] = key_type[g] = 0;
    if (g < a.n_groupby) {
      // (a reference segment over one chunk is read as the referenced segment at the PosList's offsets; PosLists over several
      //  chunks stay reference segments: hash path, row-by-row decoder)
      const DevSegment seg = resolve_segment(a.groupby[g].segments[slice.chunk], &key_pos[g]);
      if (seg.encoding == HY_ENC_REFERENCE) keys_unaligned = true;
      key_data[g] = seg.data;
      key_dictionary[g] = seg.aux;
      key_type[g] = seg.data_type;
      key_width[g] = seg.width;
      key_dictionary_size[g] = seg.aux_size;
      if (seg.encoding == HY_ENC_DICTIONARY) local_keys |= 1u << g;
      if ((seg.flags & SEG_UNALIGNED) && !key_pos[g]) keys_unaligned = true;   // (gathered value ids are read one by one: any alignment)
    }
  }
  // ---- the slice's PosList, read ONCE ---------------------------------------------------------------------------------------------------
  // The columns of a reference table share their PosList, and every column's pass over the slice used to read it again: 64 KB per slice and
  // column, with a reuse distance of all the resident workgroups' slices -- more than an L2 holds, so TPC-H Q1 behind its scan read the
  // 472 MB list five times (profiles/r06_q1_chain_launches.txt: 3.35 GB per launch).  Now the first referenced column's offsets are staged in
  // LDS (a thread reads back only what it wrote: no barrier for the data) and every column behind the same list takes them from there.
  const uint32_t* cached_pos = nullptr;
  if (a.pos_cache) {
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) if (!cached_pos && key_pos[g]) cached_pos = key_pos[g];
    for (uint32_t g = 0; g < a.n_aggregates && !cached_pos; ++g) {
      const uint32_t* pos_words = nullptr;
      if (a.aggregates[g].segments) resolve_segment(a.aggregates[g].segments[slice.chunk], &pos_words);
      cached_pos = pos_words;
    }
    if (cached_pos) {
      const global_u32* pos = reinterpret_cast<const global_u32*>(reinterpret_cast<uintptr_t>(cached_pos));
      bool encodable = true;
#pragma unroll
      for (uint32_t k = 0; k < ROWS / 4; ++k) {
        const uint32_t r0 = slice_row(k * 4, tid);
        const uint32_t n = r0 < slice.row_count ? (slice.row_count - r0 < 4 ? slice.row_count - r0 : 4u) : 0u;
        uint32_t offset[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) offset[i] = pos[2 * (size_t{slice.row_begin} + (n ? r0 + (i < n ? i : n - 1) : 0u)) + 1];   // (rows past the end: the last row's once more)
        const uint32_t d1 = offset[1] - offset[0], d2 = offset[2] - offset[1], d3 = offset[3] - offset[2];
        if (n && (offset[0] > 0xFFFFu || d1 > 31u || d2 > 31u || d3 > 31u)) encodable = false;
        s_pos[k * 256 + tid] = n ? offset[0] | d1 << 16 | d2 << 21 | d3 << 26 : 0u;
      }
      if (!encodable) s_spilled[2] = 1;
    }
    __syncthreads();
    if (s_spilled[2]) cached_pos = nullptr;
  }
  // direct-mapped table?
  uint32_t direct_size[MAX_GROUPBY], direct_stride[MAX_GROUPBY];
#pragma unroll
  for (uint32_t g = 0; g < MAX_GROUPBY; ++g) { direct_size[g] = 1; direct_stride[g] = 0; }
  bool direct = a.n_groupby > 0 && local_keys == (1u << a.n_groupby) - 1 && !keys_unaligned;   // (aligned: the wide loads of pass 1 stay inside the word that holds a chunk's last value id)
  {
    uint64_t product = 1;
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
      if (g < a.n_groupby && direct) {
        direct_size[g] = key_dictionary_size[g] + 1;   // + 1: NULL
        direct_stride[g] = static_cast<uint32_t>(product);
        product *= direct_size[g];
        if (product > LDS_SLOTS) direct = false;
      }
    }
  }
  if (stamps && tid == 0) stamps[11] = wall_clock64();
  // ---- pass 1: group lookup, 4 rows at a time -----------------------------------------------------------------------------
  constexpr int GB = 4;
  if (direct) {
    // Every GROUP BY column is a dictionary segment and the product of (dictionary size + 1) fits the table: the combined
    // value-id code IS the slot -- no hash, no probing, no key comparison, and no table traffic per row either: the rows
    // only mark their code in a presence bitmap (codes below 32 through a register first), the groups are entered after
    // the loop by the thread that owns the code.  All attribute-vector loads of the slice are in flight at once.
    if (tid < 8) s_present[tid] = 0;
    __syncthreads();
    uint32_t seen_low = 0;
    constexpr uint32_t BLOCKS = ROWS / GB;
    uint32_t codes[BLOCKS];   // the codes of a block's four rows, one byte each
#pragma unroll
    for (uint32_t block = 0; block < BLOCKS; ++block) codes[block] = 0;
    // column by column: the eight blocks' loads of a column are issued back to back, the width is decided once
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
      if (g >= a.n_groupby) continue;
      const uint32_t size = key_dictionary_size[g], stride = direct_stride[g];
      uint32_t vid[BLOCKS][GB];
      if (key_pos[g]) {   // behind a PosList: the offsets of the thread's 32 rows (one batch of loads), then their value ids (another)
        uint32_t offset[BLOCKS][GB];
        if (key_pos[g] == cached_pos) {
#pragma unroll
          for (uint32_t block = 0; block < BLOCKS; ++block) cached_offsets(s_pos[block * 256 + tid], offset[block]);
        } else {
#pragma unroll
          for (uint32_t block = 0; block < BLOCKS; ++block) {
            const uint32_t r0 = slice_row(block * GB, tid);
#pragma unroll
            for (int i = 0; i < GB; ++i) offset[block][i] = key_pos[g][2 * size_t{slice.row_begin + (r0 + i < slice.row_count ? r0 + i : 0)} + 1];
          }
        }
#pragma unroll
        for (uint32_t block = 0; block < BLOCKS; ++block) {
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            const bool null_row = offset[block][i] == 0xFFFFFFFFu;   // NULL_ROW_ID (an outer join's output): a NULL key
            const uint32_t id = aload_compressed(key_data[g], key_width[g], null_row ? 0u : offset[block][i]);
            vid[block][i] = null_row ? size : id;
          }
        }
      } else if (key_width[g] == 1) {
        const global_u32* base = reinterpret_cast<const global_u32*>(reinterpret_cast<uintptr_t>(key_data[g]));
        uint32_t word[BLOCKS];
#pragma unroll
        for (uint32_t block = 0; block < BLOCKS; ++block) {
          const uint32_t r0 = slice_row(block * GB, tid);
          word[block] = base[(slice.row_begin + (r0 < slice.row_count ? r0 : 0)) / 4];
        }
#pragma unroll
        for (uint32_t block = 0; block < BLOCKS; ++block) {
#pragma unroll
          for (int i = 0; i < GB; ++i) vid[block][i] = (word[block] >> (8 * i)) & 0xFF;
        }
      } else if (key_width[g] == 2) {
        const global_u32x2* base = reinterpret_cast<const global_u32x2*>(reinterpret_cast<uintptr_t>(key_data[g]));
        u32x2 word[BLOCKS];
#pragma unroll
        for (uint32_t block = 0; block < BLOCKS; ++block) {
          const uint32_t r0 = slice_row(block * GB, tid);
          word[block] = base[(slice.row_begin + (r0 < slice.row_count ? r0 : 0)) / 4];
        }
#pragma unroll
        for (uint32_t block = 0; block < BLOCKS; ++block) {
          vid[block][0] = word[block].x & 0xFFFF; vid[block][1] = word[block].x >> 16; vid[block][2] = word[block].y & 0xFFFF; vid[block][3] = word[block].y >> 16;
        }
      } else {
        const global_u32* base = reinterpret_cast<const global_u32*>(reinterpret_cast<uintptr_t>(key_data[g]));
#pragma unroll
        for (uint32_t block = 0; block < BLOCKS; ++block) {
          const uint32_t r0 = slice_row(block * GB, tid);
#pragma unroll
          for (int i = 0; i < GB; ++i) vid[block][i] = base[slice.row_begin + (r0 + i < slice.row_count ? r0 + i : 0)];
        }
      }
#pragma unroll
      for (uint32_t block = 0; block < BLOCKS; ++block) {
#pragma unroll
        for (int i = 0; i < GB; ++i) codes[block] += __umul24(vid[block][i] < size ? vid[block][i] : size, stride) << (8 * i);   // NULL: the last digit.  (A code is below 256: no carry; 24-bit multiply: full rate.)
      }
    }
#pragma unroll
    for (uint32_t block = 0; block < BLOCKS; ++block) {
      const uint32_t r0 = slice_row(block * GB, tid);
      const uint32_t valid = r0 + 3 < slice.row_count ? 0xFu : (r0 < slice.row_count ? (1u << (slice.row_count - r0)) - 1 : 0u);
      in_table |= valid << (block * GB);
      reinterpret_cast<uint32_t*>(s_row_slot)[block * 256 + tid] = codes[block];   // = s_row_slot[slice_row(block * 4 + i, tid)], i = 0..3
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        if (!((valid >> i) & 1)) continue;
        const uint32_t code = (codes[block] >> (8 * i)) & 0xFF;
        if (code < 32) seen_low |= 1u << code;
        else atomicOr(&s_present[code >> 5], 1u << (code & 31));
      }
    }
    {   // one LDS atomic per wave, not per thread: 256 atomics on one word are 256 serial operations
      const uint32_t wave_seen = static_cast<uint32_t>(wave_reduce_to_lane63(seen_low, 0, [](uint64_t x, uint64_t y) { return x | y; }));
      if (lane == 63 && wave_seen) atomicOr(&s_present[0], wave_seen);
    }
    __syncthreads();
    {   // thread = code: enter the group; its dense index is its rank among the codes present
      uint32_t before = 0, total = 0;
#pragma unroll
      for (uint32_t w = 0; w < 8; ++w) {
        const uint32_t bits = s_present[w];
        total += __popc(bits);
        if (w < (tid >> 5)) before += __popc(bits);
        else if (w == (tid >> 5)) before += __popc(bits & ((1u << (tid & 31)) - 1));
      }
      if (tid < LDS_SLOTS && ((s_present[tid >> 5] >> (tid & 31)) & 1)) {
        uint64_t null_mask = 0;
#pragma unroll
        for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
          if (g >= a.n_groupby) continue;
          const uint32_t digit = (tid / direct_stride[g]) % direct_size[g];
          const bool is_null = digit == direct_size[g] - 1;
          if (is_null) null_mask |= 1ull << g;
          s_keys[tid * words + g + 1] = is_null ? 0 : digit;
        }
        s_keys[tid * words] = null_mask;
        s_dense_of_slot[tid] = before;
        if (before < DENSE_GROUPS) s_slot_of_dense[before] = tid;
        s_tags[tid] = 0x80000000u;
      }
      if (tid == 0) *s_n_groups = total;
    }
  }
#pragma unroll 1
  for (uint32_t block = direct ? ROWS / GB : 0; block < ROWS / GB; ++block) {
    // A table that is full after a part of the slice: the rest of the rows would walk it in vain.  They count as rows left
    // out (pass 3 sends them to the global table -- or, when that happens in many slices, the partitioned path takes the table).
    if (__hip_atomic_load(s_n_groups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= LDS_SLOTS) break;
    uint32_t row[GB], valid = 0;
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const uint32_t r = slice_row(block * GB + i, tid);
      if (r < slice.row_count) valid |= 1u << i;
      row[i] = slice.row_begin + (r < slice.row_count ? r : 0);
    }
    // Inside a slice (one chunk) a dictionary column's VALUE ID identifies the value, so dictionary columns are keyed by
    // their value ids here and translated to values only when the slice's groups are merged into the global table:
    // pass 1 never touches the dictionaries.  The attribute-vector loads of all dictionary columns are issued first.
    uint64_t tuple[GB][MAX_GROUPBY + 1];
#pragma unroll
    for (int i = 0; i < GB; ++i) tuple[i][0] = 0;
    uint32_t vid[MAX_GROUPBY][GB];
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
      if (g >= a.n_groupby || !((local_keys >> g) & 1)) continue;
      // the block's four rows are consecutive and start at a multiple of four: one aligned load of all four value ids
      // (not for the group that straddles the end of the chunk: nothing may be read behind a caller's buffer)
      if (key_pos[g]) {
        uint32_t offset[GB];
        if (key_pos[g] == cached_pos) {
          cached_offsets(s_pos[block * 256 + tid], offset);
        } else {
#pragma unroll
          for (int i = 0; i < GB; ++i) offset[i] = key_pos[g][2 * size_t{row[i]} + 1];
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) vid[g][i] = offset[i] == 0xFFFFFFFFu ? key_dictionary_size[g] : aload_compressed(key_data[g], key_width[g], offset[i]);
      } else if (key_width[g] == 1 && valid == 0xF) {
        const uint32_t four = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(key_data[g]) + row[0]);
#pragma unroll
        for (int i = 0; i < GB; ++i) vid[g][i] = (four >> (8 * i)) & 0xFF;
      } else if (key_width[g] == 2 && valid == 0xF) {
        const u32x2 four = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(key_data[g]) + row[0]);
        vid[g][0] = four.x & 0xFFFF; vid[g][1] = four.x >> 16; vid[g][2] = four.y & 0xFFFF; vid[g][3] = four.y >> 16;
      } else {
#pragma unroll
        for (int i = 0; i < GB; ++i) vid[g][i] = aload_compressed(key_data[g], key_width[g], row[i]);
      }
    }
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
      if (g >= a.n_groupby) {
#pragma unroll
        for (int i = 0; i < GB; ++i) tuple[i][g + 1] = 0;
        continue;
      }
      if ((local_keys >> g) & 1) {
        const uint32_t dictionary_size = key_dictionary_size[g];
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          const bool is_null = vid[g][i] >= dictionary_size;
          if (is_null) tuple[i][0] |= 1ull << g;
          tuple[i][g + 1] = is_null ? 0 : vid[g][i];
        }
        continue;
      }
      uint64_t bits[GB];
      uint32_t nulls;
      {
        const uint32_t* pos_words;
        const DevSegment key_segment = resolve_segment(a.groupby[g].segments[slice.chunk], &pos_words);
        uint32_t key_row[GB];
#pragma unroll
        for (int i = 0; i < GB; ++i) key_row[i] = row[i];
        const uint32_t null_rows = pos_words ? dereference_rows<GB>(pos_words, key_row) : 0u;
        decode_rows<GB>(key_segment, a.groupby[g].segments, slice.chunk, key_row, valid, bits, &nulls);
        nulls |= null_rows;
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        uint64_t word = bits[i];
        if (a.groupby[g].is_float && __longlong_as_double(static_cast<long long>(word)) == 0.0) word = 0;   // -0.0 groups with 0.0
        if ((nulls >> i) & 1) { tuple[i][0] |= 1ull << g; word = 0; }
        tuple[i][g + 1] = word;
      }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      if (!((valid >> i) & 1)) continue;
      const uint32_t k = block * GB + i;
      const uint64_t hash = hash_local(tuple[i], words);
      // workgroup-private table in LDS (same lock discipline as global_slot)
      const uint32_t ready = 0x80000000u | static_cast<uint32_t>(hash >> 33);
      uint32_t slot = static_cast<uint32_t>(hash) & (LDS_SLOTS - 1);
      bool found = false;
      uint32_t probes = 0;
      bool done = false;
      while (!done) {
        // A plain ds_read (256 threads reading 4 hot tags with an atomic RMW serialise), relaxed: LDS operations of a wave
        // execute in order, so the key words read below are the ones written before the tag was published; the signal
        // fence keeps the compiler from moving those reads up.  (An acquire load waits for every outstanding load.)
        const uint32_t tag = __hip_atomic_load(&s_tags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __atomic_signal_fence(__ATOMIC_ACQUIRE);
        if (tag == TAG_EMPTY) {
          if (atomicCAS(&s_tags[slot], TAG_EMPTY, TAG_LOCKED) == TAG_EMPTY) {
#pragma unroll
            for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) { if (w < words) s_keys[slot * words + w] = tuple[i][w]; }
            const uint32_t dense = atomicAdd(s_n_groups, 1u);
            s_dense_of_slot[slot] = dense;
            if (dense < DENSE_GROUPS) s_slot_of_dense[dense] = slot;
            __threadfence_block();
            atomicExch(&s_tags[slot], ready);
            found = true;
            done = true;
          }
        } else if (tag != TAG_LOCKED) {
          bool equal = tag == ready;
          if (equal) {
#pragma unroll
            for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) { if (w < words) equal &= s_keys[slot * words + w] == tuple[i][w]; }
          }
          if (equal) {
            found = true;
            done = true;
          } else {
            slot = (slot + 1) & (LDS_SLOTS - 1);
            if (++probes >= MAX_LDS_PROBES) done = true;   // a crowded (or full) LDS table: this row goes to the global table -- wherever a group's rows
                                                           // accumulate, they meet in the global table under the group's key
          }
        }
      }
      if (found) in_table |= 1u << k;
      s_row_slot[slice_row(k, tid)] = static_cast<uint8_t>(slot);
    }
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[1] = wall_clock64();
  const uint32_t n_groups = *s_n_groups;
  const uint32_t n_dense = n_groups < DENSE_GROUPS ? n_groups : DENSE_GROUPS;
  if (n_groups > LDS_SLOTS / 2) {   // a crowded LDS table: rows may have been left out.  Many of them in many slices = a table for the partitioned path
    uint32_t outside = 0;
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) outside += (slice_row(k, tid) < slice.row_count && !((in_table >> k) & 1)) ? 1u : 0u;
    const uint32_t wave_outside = wave_reduce_u32_to_lane63(outside, 0u, false, false);
    if (lane == 63 && wave_outside) atomicAdd(&s_spilled[0], wave_outside);
    __syncthreads();
    if (tid == 0) {
      if (s_spilled[0]) {
        // Two ways to learn that the table has too many groups for this kernel: slices most of whose rows are outside their LDS table
        // (thousands of groups: the first few such slices say so before any of them has paid for its rows -- waiting for a share of ALL
        // rows to spill made the abandoned attempt cost more than the partitioned path that follows), and the sum of the rows outside
        // (a few groups more than the table holds, SSB Q2.1's 280: those rows take device-scope atomics, up to the limit).
        const uint32_t before = atomicAdd(&a.overflow[FLAG_SPILLED], s_spilled[0]);
        bool give_up = before + s_spilled[0] > a.spill_limit;
        if (a.spill_limit != 0xFFFFFFFFu && 2 * s_spilled[0] > slice.row_count) give_up |= atomicAdd(&a.overflow[FLAG_CROWDED], 1u) >= CROWDED_SLICES;
        if (give_up) __hip_atomic_store(&a.overflow[FLAG_GIVE_UP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_spilled[1] = __hip_atomic_load(&a.overflow[FLAG_GIVE_UP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_spilled[1]) return;   // the whole attempt is being abandoned: nothing this slice adds will be read
  }

  // ---- pass 2: rows of the dense groups, registers --------------------------------------------------------------------------
  uint32_t dense_lo = 0, dense_hi = 0;   // dense index of every row, two bits per row (rows 0-15 | 16-31); 3 = not dense when n_dense < 4 ...
  uint32_t is_dense = 0;                 // ... so the membership is kept separately
  {
    uint32_t slots[ROWS / 4];
#pragma unroll
    for (uint32_t m = 0; m < ROWS / 4; ++m) slots[m] = reinterpret_cast<const uint32_t*>(s_row_slot)[m * 256 + tid];   // rows 4m .. 4m+3 of the thread
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) {
      const uint32_t r = slice_row(k, tid);
      if (r < slice.row_count && ((in_table >> k) & 1)) {
        const uint32_t dense = s_dense_of_slot[(slots[k / 4] >> (8 * (k & 3))) & 0xFF];   // (a table lookup: computing the rank of the code from the presence bits costs more)
        if (dense < DENSE_GROUPS) {
          is_dense |= 1u << k;
          if (k < 16) dense_lo |= dense << (2 * k); else dense_hi |= dense << (2 * (k - 16));
        }
      }
    }
    // A thread's rows ascend with k: the first / last row of a group is the lowest / highest set bit of the group's
    // membership mask (the two-bit codes of 16 rows compared at once, the even bits of the match squeezed together).
    uint32_t first[DENSE_GROUPS], last[DENSE_GROUPS];
    auto squeeze = [](uint32_t x) {   // bits 0, 2, 4 ... 30 -> bits 0 .. 15
      x &= 0x55555555u;
      x = (x | x >> 1) & 0x33333333u;
      x = (x | x >> 2) & 0x0F0F0F0Fu;
      x = (x | x >> 4) & 0x00FF00FFu;
      return (x | x >> 8) & 0xFFFFu;
    };
#pragma unroll
    for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {
      const uint32_t want_low = (j & 1) ? 0xFFFFFFFFu : 0u, want_high = (j & 2) ? 0xFFFFFFFFu : 0u;
      const uint32_t match_lo = ~(dense_lo ^ want_low) & ~((dense_lo >> 1) ^ want_high), match_hi = ~(dense_hi ^ want_low) & ~((dense_hi >> 1) ^ want_high);
      const uint32_t members = (squeeze(match_lo) | squeeze(match_hi) << 16) & is_dense;
      first[j] = members ? slice_row(static_cast<uint32_t>(__ffs(static_cast<int>(members))) - 1, tid) : 0xFFFFFFFFu;
      last[j] = members ? slice_row(31u - static_cast<uint32_t>(__clz(static_cast<int>(members))), tid) : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {   // representative rows: smallest / largest row of the group
      if (j >= n_dense) break;
      const uint32_t lowest = wave_reduce_u32_to_lane63(first[j], 0xFFFFFFFFu, true, false), highest = wave_reduce_u32_to_lane63(last[j], 0u, false, true);
      if (lane == 63 && lowest != 0xFFFFFFFFu) {
        const uint32_t slot = s_slot_of_dense[j];
        atomicMin(reinterpret_cast<unsigned long long*>(&s_first[slot]), static_cast<unsigned long long>(chunk_base + slice.row_begin + lowest));
        atomicMax(reinterpret_cast<unsigned long long*>(&s_last[slot]), static_cast<unsigned long long>(chunk_base + slice.row_begin + highest));
      }
    }
  }
  constexpr int AB = 16;
  if (stamps && tid == 0) stamps[2] = wall_clock64();
#pragma unroll 1
  for (uint32_t g = 0; g < a.n_aggregates; ++g) {
    if (stamps && tid == 0 && g < 6) stamps[3 + g] = wall_clock64();
    const AggColumn c = a.aggregates[g];
    const uint32_t* value_pos_words = nullptr;   // the slice's rows sit behind a single-chunk PosList: their offsets in the referenced segment
    const DevSegment value_segment = c.segments ? resolve_segment(c.segments[slice.chunk], &value_pos_words) : DevSegment{};
    // what the accumulators do, decided once per aggregate (not per row)
    enum : uint32_t { ACC_NONE, ACC_MIN, ACC_MAX, ACC_ADD_INT, ACC_ADD_DOUBLE };
    uint32_t kind = ACC_NONE;
    if (c.function == HY_AGG_MIN) kind = ACC_MIN;
    else if (c.function == HY_AGG_MAX) kind = ACC_MAX;
    else if (c.function == HY_AGG_SUM) kind = c.is_float ? ACC_ADD_DOUBLE : ACC_ADD_INT;
    else if (c.function == HY_AGG_AVG || c.function == AGG_SUM_SQUARES || c.function == AGG_SUM_SHIFTED) kind = ACC_ADD_DOUBLE;
    const bool to_ordered = (kind == ACC_MIN || kind == ACC_MAX) && c.is_float;   // MIN/MAX of doubles on order-preserving int64
    const bool int_to_double = c.function == HY_AGG_AVG && !c.is_float;
    // One private accumulator per (dense group, thread), in registers: a row updates the accumulator of its group through
    // selects (for every group: combine, keep the old value unless the row belongs to it).  That is four combines per row
    // instead of one, and still several times faster than one LDS atomic per row into a private cell: ds_add_f64 retires
    // about one lane per cycle.
    // The usual aggregate column -- an unencoded, aligned 4-byte segment without NULLs -- skips the generic decoder.
    const bool plain4 = c.segments && !value_pos_words && value_segment.encoding == HY_ENC_UNENCODED && !(value_segment.flags & SEG_UNALIGNED) &&   // (with or without a null bitmap: a projection's result always has one)
                        (value_segment.data_type == HY_TYPE_INT || value_segment.data_type == HY_TYPE_FLOAT);
    uint64_t cell_value[DENSE_GROUPS];
    uint32_t cell_count[DENSE_GROUPS];   // the thread's non-NULL rows of every dense group: population counts
#pragma unroll
    for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {
      cell_value[j] = initial_value(c.function);
      cell_count[j] = 0;
    }
#pragma unroll 1
    for (uint32_t half = 0; half < ROWS / AB; ++half) {
      const uint32_t members = (is_dense >> (half * AB)) & 0xFFFFu;
      const uint32_t dense_bits = half == 0 ? dense_lo : dense_hi;
      uint64_t bits[AB];
      uint32_t nulls = 0;
      if (c.segments) {
        uint32_t row[AB];
#pragma unroll
        for (int i = 0; i < AB; ++i) {
          const uint32_t r = slice_row(half * AB + i, tid);
          row[i] = slice.row_begin + (r < slice.row_count ? r : 0);
        }
        if (plain4) {   // four aligned 16-byte loads instead of sixteen decoded rows
          const global_u32x4* base = reinterpret_cast<const global_u32x4*>(reinterpret_cast<uintptr_t>(value_segment.data));
          u32x4 raw[AB / 4];
#pragma unroll
          for (int m = 0; m < AB / 4; ++m) raw[m] = base[row[4 * m] / 4];   // (rows past the slice's end were clamped to row 0 of the slice: their values are not taken)
          if (value_segment.nulls) {   // the four rows' bits: a nibble of one byte of the bitmap (row[4 m] is a multiple of four)
            typedef __attribute__((address_space(1))) const uint8_t global_byte;
            global_byte* null_bytes = reinterpret_cast<global_byte*>(reinterpret_cast<uintptr_t>(value_segment.nulls));
            uint32_t null_byte[AB / 4];
#pragma unroll
            for (int m = 0; m < AB / 4; ++m) null_byte[m] = null_bytes[row[4 * m] / 8];
#pragma unroll
            for (int m = 0; m < AB / 4; ++m) nulls |= ((null_byte[m] >> (row[4 * m] & 4u)) & 0xFu) << (4 * m);
          }
          const bool is_float = value_segment.data_type == HY_TYPE_FLOAT;
#pragma unroll
          for (int i = 0; i < AB; ++i) {
            const uint32_t word = raw[i / 4][i & 3];
            bits[i] = is_float ? static_cast<uint64_t>(__double_as_longlong(static_cast<double>(__uint_as_float(word)))) : static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(word)));
          }
        } else {
          uint32_t null_rows = 0;
          if (value_pos_words && value_pos_words == cached_pos) {
#pragma unroll
            for (int m = 0; m < AB / 4; ++m) {
              uint32_t offset[4];
              cached_offsets(s_pos[(half * (AB / 4) + m) * 256 + tid], offset);
#pragma unroll
              for (int i = 0; i < 4; ++i) row[4 * m + i] = offset[i];
            }
          } else if (value_pos_words) {
            null_rows = dereference_rows<AB>(value_pos_words, row);
          }
          decode_rows<AB>(value_segment, c.segments, slice.chunk, row, members, bits, &nulls);
          nulls |= null_rows;
        }
        if (to_ordered) {
#pragma unroll
          for (int i = 0; i < AB; ++i) bits[i] = static_cast<uint64_t>(ordered_bits(__longlong_as_double(static_cast<long long>(bits[i]))));
        } else if (int_to_double) {
#pragma unroll
          for (int i = 0; i < AB; ++i) bits[i] = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<int64_t>(bits[i]))));
        } else if (c.function == AGG_SUM_SQUARES || c.function == AGG_SUM_SHIFTED) {
#pragma unroll
          for (int i = 0; i < AB; ++i) bits[i] = contribution_from(c, bits[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < AB; ++i) bits[i] = 0;   // COUNT(*)
      }
      const uint32_t take = members & ~nulls;
      {   // rows taken per dense group: `take` spread to the even bit positions, against the group's two-bit codes
        uint32_t spread = take;
        spread = (spread | spread << 8) & 0x00FF00FFu;
        spread = (spread | spread << 4) & 0x0F0F0F0Fu;
        spread = (spread | spread << 2) & 0x33333333u;
        spread = (spread | spread << 1) & 0x55555555u;
        const uint32_t low = dense_bits & 0x55555555u, high = (dense_bits >> 1) & 0x55555555u;
        cell_count[0] += __popc(spread & ~low & ~high);
        cell_count[1] += __popc(spread & low & ~high);
        cell_count[2] += __popc(spread & ~low & high);
        cell_count[3] += __popc(spread & low & high);
      }
      // the rows of group j among the sixteen: bit i of mine[j]
      uint32_t mine[DENSE_GROUPS];
      {
        auto squeeze = [](uint32_t x) {   // bits 0, 2, 4 ... 30 -> bits 0 .. 15
          x = (x | x >> 1) & 0x33333333u;
          x = (x | x >> 2) & 0x0F0F0F0Fu;
          x = (x | x >> 4) & 0x00FF00FFu;
          return (x | x >> 8) & 0xFFFFu;
        };
        const uint32_t low = dense_bits & 0x55555555u, high = (dense_bits >> 1) & 0x55555555u;
        mine[0] = squeeze(~low & ~high & 0x55555555u) & take;
        mine[1] = squeeze(low & ~high) & take;
        mine[2] = squeeze(~low & high) & take;
        mine[3] = squeeze(low & high) & take;
      }
      switch (kind) {   // one loop per kind: the row loop itself stays free of scalar branches
        case ACC_MIN:
#pragma unroll
          for (int i = 0; i < AB; ++i) {
#pragma unroll
            for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {
              const long long smaller = min(static_cast<long long>(cell_value[j]), static_cast<long long>(bits[i]));
              cell_value[j] = ((mine[j] >> i) & 1) ? static_cast<uint64_t>(smaller) : cell_value[j];
            }
          }
          break;
        case ACC_MAX:
#pragma unroll
          for (int i = 0; i < AB; ++i) {
#pragma unroll
            for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {
              const long long larger = max(static_cast<long long>(cell_value[j]), static_cast<long long>(bits[i]));
              cell_value[j] = ((mine[j] >> i) & 1) ? static_cast<uint64_t>(larger) : cell_value[j];
            }
          }
          break;
        case ACC_ADD_INT:
#pragma unroll
          for (int i = 0; i < AB; ++i) {
#pragma unroll
            for (uint32_t j = 0; j < DENSE_GROUPS; ++j) cell_value[j] += ((mine[j] >> i) & 1) ? bits[i] : 0ull;
          }
          break;
        case ACC_ADD_DOUBLE:
#pragma unroll
          for (int i = 0; i < AB; ++i) {
#pragma unroll
            for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {   // (+ 0.0 for the other groups' rows: exact)
              const double addend = ((mine[j] >> i) & 1) ? __longlong_as_double(static_cast<long long>(bits[i])) : 0.0;
              cell_value[j] = static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(cell_value[j])) + addend));
            }
          }
          break;
        default: break;
      }
    }
    if (stamps && tid == 0 && g == 0 && a.n_aggregates <= 5) stamps[8] = wall_clock64();   // (debug: end of the first accumulator's row loop)
    // the wave's totals (lane 63): the combine is chosen ONCE per accumulator -- inside the DPP steps, a switch on the
    // aggregate function was a dozen scalar branches per step, 48 steps per accumulator: more time than the rows took
    uint64_t reduced[DENSE_GROUPS];
    {
      const uint64_t identity = initial_value(c.function);
      switch (kind) {
        case ACC_MIN:
#pragma unroll
          for (uint32_t j = 0; j < DENSE_GROUPS; ++j)
            reduced[j] = wave_reduce_to_lane63(cell_value[j], identity, [](uint64_t x, uint64_t y) { return static_cast<uint64_t>(min(static_cast<long long>(x), static_cast<long long>(y))); });
          break;
        case ACC_MAX:
#pragma unroll
          for (uint32_t j = 0; j < DENSE_GROUPS; ++j)
            reduced[j] = wave_reduce_to_lane63(cell_value[j], identity, [](uint64_t x, uint64_t y) { return static_cast<uint64_t>(max(static_cast<long long>(x), static_cast<long long>(y))); });
          break;
        case ACC_ADD_INT:
#pragma unroll
          for (uint32_t j = 0; j < DENSE_GROUPS; ++j) reduced[j] = wave_reduce_to_lane63(cell_value[j], identity, [](uint64_t x, uint64_t y) { return x + y; });
          break;
        case ACC_ADD_DOUBLE:
#pragma unroll
          for (uint32_t j = 0; j < DENSE_GROUPS; ++j)
            reduced[j] = wave_reduce_to_lane63(cell_value[j], identity, [](uint64_t x, uint64_t y) {
              return static_cast<uint64_t>(__double_as_longlong(__longlong_as_double(static_cast<long long>(x)) + __longlong_as_double(static_cast<long long>(y))));
            });
          break;
        default:
#pragma unroll
          for (uint32_t j = 0; j < DENSE_GROUPS; ++j) reduced[j] = cell_value[j];
          break;
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < DENSE_GROUPS; ++j) {
      if (j >= n_dense) break;
      const uint64_t value = reduced[j];
      const uint32_t count = wave_reduce_u32_to_lane63(cell_count[j], 0u, false, false);
      if (lane == 63 && count != 0) {
        const uint32_t slot = s_slot_of_dense[j];
        accumulate_lds(c, &s_values[slot * a.n_aggregates + g], value);
        atomicAdd(&s_counts[slot * a.n_aggregates + g], count);
      }
    }
  }

  if (stamps && tid == 0) stamps[9] = wall_clock64();
  // ---- pass 3: rows of the other groups (slices with more than DENSE_GROUPS groups only) ----------------------------------
  if (n_groups > DENSE_GROUPS) {
#pragma unroll 1
    for (uint32_t k = 0; k < ROWS; ++k) {
      const uint32_t r = slice_row(k, tid);
      if (r >= slice.row_count || ((is_dense >> k) & 1)) continue;
      if (__hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // the table is too small: the host starts over with a larger one, no point in walking the full one
      if (!((in_table >> k) & 1) && __hip_atomic_load(&a.overflow[FLAG_GIVE_UP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // ... or with the partitioned path
      const uint32_t row = slice.row_begin + r;
      const uint64_t global_row = chunk_base + row;
      const bool found = (in_table >> k) & 1;
      const uint32_t slot = s_row_slot[slice_row(k, tid)];
      uint32_t gslot = 0xFFFFFFFFu;
      if (!found) {
        uint64_t tuple[MAX_GROUPBY + 1];
        row_tuple(a, slice.chunk, row, tuple);
        gslot = global_slot(a, tuple, words, hash_tuple(tuple, words));
        if (gslot == 0xFFFFFFFFu) { *a.overflow = 1; continue; }
        atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(global_row));
        atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(global_row));
      } else {
        atomicMin(reinterpret_cast<unsigned long long*>(&s_first[slot]), static_cast<unsigned long long>(global_row));
        atomicMax(reinterpret_cast<unsigned long long*>(&s_last[slot]), static_cast<unsigned long long>(global_row));
      }
      for (uint32_t g = 0; g < a.n_aggregates; ++g) {
        const AggColumn& c = a.aggregates[g];
        uint64_t contribution;
        if (!contribution_of(c, slice.chunk, row, &contribution)) continue;
        if (!found) { merge_global(a, gslot, g, contribution, 1); continue; }
        accumulate_lds(c, &s_values[slot * a.n_aggregates + g], contribution);
        atomicAdd(&s_counts[slot * a.n_aggregates + g], 1u);
      }
    }
  }
  __syncthreads();
  // merge the workgroup's groups into the global table
  for (uint32_t s = tid; s < LDS_SLOTS; s += 256) {
    if (s_tags[s] == TAG_EMPTY) continue;
    if (__hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    uint64_t tuple[MAX_GROUPBY + 1];
    for (uint32_t w = 0; w < words; ++w) tuple[w] = s_keys[s * words + w];
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {   // value ids -> values (the dictionary pointers were read with the other descriptor fields)
      if (g >= a.n_groupby || !((local_keys >> g) & 1) || ((tuple[0] >> g) & 1)) continue;
      const uint32_t id = static_cast<uint32_t>(tuple[g + 1]);
      uint64_t bits;
      switch (key_type[g]) {
        case HY_TYPE_INT: bits = static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int32_t*>(key_dictionary[g])[id])); break;
        case HY_TYPE_LONG: bits = static_cast<const uint64_t*>(key_dictionary[g])[id]; break;
        case HY_TYPE_FLOAT: bits = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(static_cast<const float*>(key_dictionary[g])[id]))); break;
        default: bits = static_cast<const uint64_t*>(key_dictionary[g])[id]; break;
      }
      if (a.groupby[g].is_float && __longlong_as_double(static_cast<long long>(bits)) == 0.0) bits = 0;
      tuple[g + 1] = bits;
    }
    const uint32_t gslot = global_slot(a, tuple, words, hash_tuple(tuple, words));
    if (gslot == 0xFFFFFFFFu) { *a.overflow = 1; continue; }
    atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(s_first[s]));
    atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(s_last[s]));
    for (uint32_t g = 0; g < a.n_aggregates; ++g) merge_global(a, gslot, g, s_values[s * a.n_aggregates + g], s_counts[s * a.n_aggregates + g]);
  }
  if (stamps) {
    __syncthreads();
    if (tid == 0) stamps[10] = wall_clock64();
  }
}

#include "aggregate_small.hpp"

// ---- the partitioned path: tables with more groups than a slice's LDS table holds -------------------------------------------
// aggregate_rows sends the rows of groups that do not fit its 256-slot table to the global table one device-scope atomic at a
// time (a dozen G atomics/s for the whole device: 22 ms for 60 M rows in 1000 groups).  When that happens to more than a few
// rows, the host starts over here (the reference partitions by the hash of the keys for the same reason, aggregate_hash.cpp:661-
// 948: its partitions keep the hash tables in cache):
//   partition_count    per tile (one slice, or eight where there are very many partitions): histogram of the rows' partitions -- the
//                      top bits of the tuple hash
//   scan_*             exclusive prefix sum over (partition, tile): where every tile's rows of every partition go
//   partition_scatter  one RECORD per row, grouped by partition: global row number + which aggregate inputs are non-NULL | the GROUP BY
//                      tuple | the aggregates' contributions -- everything the next kernel needs, written and read back
//                      sequentially.  (Carrying RowIDs and gathering the values again costs a 128-byte line per row and column.)
//   aggregate_partitions  one workgroup per partition: ALL rows of a group are here, so the groups live in an LDS table for the
//                      whole pass (LDS atomics per row) and reach the global table once, at the end.
// The groups of the table, densely.  The first `staged_capacity` of them also go straight into pinned host memory (same
// five arrays, `staged_capacity` rows each, behind a 64-byte header): few groups -- the usual case -- cost no copy at all.
struct StagedGroups {
  uint64_t* keys;
  uint64_t* first;
  uint64_t* last;
  uint64_t* values;
  uint64_t* counts;
  uint32_t capacity;
};
// aggregate_partitions with one workgroup per partition owns its groups: they go from its LDS table straight into the compacted arrays (a range
// reserved with one atomic on the group counter) -- not group by group through the global table: 4 M groups were 28 M scattered device-scope
// atomics into a 2 GB table, 9 of the call's 14 ms on the device.  (Rows that did not fit the LDS table still meet in the global table;
// compact_groups appends its groups behind these.)
struct DirectGroups {
  uint32_t enabled, out_capacity;
  uint32_t* counter;
  uint64_t *keys, *first, *last, *values, *counts;
  StagedGroups staged;
};
struct PartitionArgs {
  uint32_t tile_slices;     // a tile of the partitioning kernels: this many consecutive slices (one workgroup)
  uint32_t n_slices;
  uint32_t reserved;
  uint32_t n_parts;         // number of tiles
  uint32_t bits;            // partitions = 1 << bits
  uint32_t lds_slots;       // aggregate_partitions: slots of the workgroup's table (power of two)
  uint32_t split;           // aggregate_partitions: workgroups per partition (each takes a share of its rows; they meet in the global table)
  uint32_t* offsets;        // [partitions][n_parts]: counts, then (after the scan) first output position
  uint64_t* records;        // [total_rows][record_words]: row | present << 32, tuple[words], contribution of every aggregate with a column
  uint64_t total_rows;
  uint32_t narrow;          // 1: NARROW records (below)
  uint32_t record_words;    // even: records are written and read as 16-byte pairs
  uint32_t carried;         // aggregates with a column (the first `carried` of AggArgs.aggregates): their contributions travel in the record
};

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t MAX_RECORD_PAIRS = (2 + MAX_GROUPBY + 1 + MAX_AGGREGATES) / 2;   // head + tuple + contributions, rounded up to 16 bytes

// NARROW records: every GROUP BY column and every carried aggregate input is a 4-byte type (int32 / float32 -- order keys, part keys, dates,
// prices): the record is 32-bit words -- row | present + NULL mask << 16 | the keys' values | the inputs' values -- padded to 16 bytes.  One int32
// key and one float input: 16 bytes a row instead of 32; the two kernels that write and read the records move half the bytes.
constexpr uint32_t MAX_NARROW_QUADS = (2 + MAX_GROUPBY + MAX_AGGREGATES + 3) / 4;
__device__ __forceinline__ uint32_t narrow_word(uint64_t bits, bool is_float) {   // the decoded word of a 4-byte column (aggregate_contribution)
  return is_float ? __float_as_uint(static_cast<float>(__longlong_as_double(static_cast<long long>(bits)))) : static_cast<uint32_t>(bits);
}
__device__ __forceinline__ uint64_t widen_word(uint32_t word, bool is_float) {
  return is_float ? static_cast<uint64_t>(__double_as_longlong(static_cast<double>(__uint_as_float(word)))) : static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(word)));
}

// WORDS = GROUP BY columns + 1 (the tuple's words): the record layout is static per instantiation, so a record is built in
// registers and leaves as 16-byte stores (8-byte stores retire at half the rate).
template <bool SCATTER, int WORDS, bool NARROW>
__global__ __launch_bounds__(256) void partition_rows(AggArgs a, PartitionArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* s_cell = reinterpret_cast<uint32_t*>(smem);   // count (SCATTER: next output position) of every partition
  const uint32_t partitions = 1u << p.bits, tid = threadIdx.x;
  for (uint32_t i = tid; i < partitions; i += 256) s_cell[i] = SCATTER ? p.offsets[size_t{i} * p.n_parts + blockIdx.x] : 0u;
  __syncthreads();
  const uint32_t pairs = p.record_words / 2, carried = p.carried;
  const uint32_t first_slice = blockIdx.x * p.tile_slices, last_slice = min(p.n_slices, first_slice + p.tile_slices);
  constexpr int R = 2;   // rows per thread and step: the loads of both (and, unrolled, of the next step) are in flight together
  for (uint32_t s = first_slice; s < last_slice; ++s) {
    const Slice slice = a.slices[s];
    const uint64_t chunk_base = SCATTER ? a.row_base[slice.chunk] : 0;
#pragma unroll 2
    for (uint32_t base = 0; base < slice.row_count; base += 256 * R) {
      uint32_t row[R], valid = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const uint32_t r = base + tid * R + i;
        if (r < slice.row_count) valid |= 1u << i;
        row[i] = slice.row_begin + (r < slice.row_count ? r : 0);
      }
      uint64_t tuple[R][MAX_GROUPBY + 1];
#pragma unroll
      for (int i = 0; i < R; ++i) {
#pragma unroll
        for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) tuple[i][w] = 0;
      }
#pragma unroll
      for (int g = 0; g < WORDS - 1; ++g) {
        const uint32_t* pos_words;
        const DevSegment segment = resolve_segment(a.groupby[g].segments[slice.chunk], &pos_words);
        uint32_t key_row[R];
#pragma unroll
        for (int i = 0; i < R; ++i) key_row[i] = row[i];
        const uint32_t null_rows = pos_words ? dereference_rows<R>(pos_words, key_row) : 0u;
        uint64_t bits[R];
        uint32_t nulls;
        decode_rows<R>(segment, a.groupby[g].segments, slice.chunk, key_row, valid, bits, &nulls);
        nulls |= null_rows;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          uint64_t word = bits[i];
          if (a.groupby[g].is_float && __longlong_as_double(static_cast<long long>(word)) == 0.0) word = 0;   // -0.0 groups with 0.0
          if ((nulls >> i) & 1) { tuple[i][0] |= 1ull << g; word = 0; }
          tuple[i][g + 1] = word;
        }
      }
      uint32_t position[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        position[i] = 0;
        if ((valid >> i) & 1) position[i] = atomicAdd(&s_cell[static_cast<uint32_t>(hash_tuple(tuple[i], WORDS) >> (64 - p.bits))], 1u);
      }
      if (SCATTER) {
        uint64_t contribution[MAX_AGGREGATES][R];
        uint32_t present[R];
#pragma unroll
        for (int i = 0; i < R; ++i) present[i] = 0;
#pragma unroll
        for (uint32_t g = 0; g < MAX_AGGREGATES; ++g) {   // the aggregates with a column come first (run_aggregate orders them)
#pragma unroll
          for (int i = 0; i < R; ++i) contribution[g][i] = 0;
          if (g >= carried) continue;
          const AggColumn c = a.aggregates[g];
          const uint32_t* pos_words;
          const DevSegment segment = resolve_segment(c.segments[slice.chunk], &pos_words);
          uint32_t value_row[R];
#pragma unroll
          for (int i = 0; i < R; ++i) value_row[i] = row[i];
          const uint32_t null_rows = pos_words ? dereference_rows<R>(pos_words, value_row) : 0u;
          uint64_t bits[R];
          uint32_t nulls;
          decode_rows<R>(segment, c.segments, slice.chunk, value_row, valid, bits, &nulls);
          nulls |= null_rows;
#pragma unroll
          for (int i = 0; i < R; ++i) {
            contribution[g][i] = NARROW ? static_cast<uint64_t>(narrow_word(bits[i], c.is_float)) : contribution_from(c, bits[i]);
            if (!((nulls >> i) & 1)) present[i] |= 1u << g;
          }
        }
        if (NARROW) {
#pragma unroll
          for (int i = 0; i < R; ++i) {
            if (!((valid >> i) & 1)) continue;
            uint32_t record[4 * MAX_NARROW_QUADS];
#pragma unroll
            for (uint32_t k = 0; k < 4 * MAX_NARROW_QUADS; ++k) record[k] = 0;
            record[0] = static_cast<uint32_t>(chunk_base + row[i]);
            record[1] = present[i] | static_cast<uint32_t>(tuple[i][0]) << 16;
#pragma unroll
            for (int w = 1; w < WORDS; ++w) record[1 + w] = narrow_word(tuple[i][w], a.groupby[w - 1].is_float);
#pragma unroll
            for (uint32_t g = 0; g < MAX_AGGREGATES; ++g) record[1 + WORDS + g] = static_cast<uint32_t>(contribution[g][i]);
            u32x4* out = reinterpret_cast<u32x4*>(p.records) + size_t{position[i]} * pairs;
#pragma unroll
            for (uint32_t k = 0; k < MAX_NARROW_QUADS; ++k) {
              if (k < pairs) { u32x4 v; v.x = record[4 * k]; v.y = record[4 * k + 1]; v.z = record[4 * k + 2]; v.w = record[4 * k + 3]; out[k] = v; }
            }
          }
          continue;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
          if (!((valid >> i) & 1)) continue;
          uint64_t record[2 * MAX_RECORD_PAIRS];
#pragma unroll
          for (uint32_t k = 0; k < 2 * MAX_RECORD_PAIRS; ++k) record[k] = 0;
          record[0] = (chunk_base + row[i]) | static_cast<uint64_t>(present[i]) << 32;
#pragma unroll
          for (int w = 0; w < WORDS; ++w) record[1 + w] = tuple[i][w];
#pragma unroll
          for (uint32_t g = 0; g < MAX_AGGREGATES; ++g) record[1 + WORDS + g] = contribution[g][i];
          u64x2* out = reinterpret_cast<u64x2*>(p.records) + size_t{position[i]} * pairs;
#pragma unroll
          for (uint32_t k = 0; k < MAX_RECORD_PAIRS; ++k) {
            if (k < pairs) { u64x2 v; v.x = record[2 * k]; v.y = record[2 * k + 1]; out[k] = v; }
          }
        }
      }
    }
  }
  if (SCATTER) return;
  __syncthreads();
  for (uint32_t i = tid; i < partitions; i += 256) p.offsets[size_t{i} * p.n_parts + blockIdx.x] = s_cell[i];
}

// Exclusive prefix sum of `data[0 .. n)` in place, three launches: per block of 4096 (sums[b] = the block's total), the block
// totals (one workgroup; sums[n_blocks] = the grand total), and the block offsets added back.
constexpr uint32_t SCAN_BLOCK = 4096;
__global__ __launch_bounds__(256) void scan_blocks(uint32_t* data, uint64_t n, uint32_t* sums) {
  __shared__ uint32_t s_total[256];
  const uint64_t begin = uint64_t{blockIdx.x} * SCAN_BLOCK + threadIdx.x * 16;
  uint32_t v[16], total = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = begin + i < n ? data[begin + i] : 0u; total += v[i]; }
  s_total[threadIdx.x] = total;
  __syncthreads();
  for (uint32_t step = 1; step < 256; step <<= 1) {   // Hillis-Steele over the 256 thread totals
    const uint32_t add = threadIdx.x >= step ? s_total[threadIdx.x - step] : 0u;
    __syncthreads();
    s_total[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t running = threadIdx.x ? s_total[threadIdx.x - 1] : 0u;
#pragma unroll
  for (int i = 0; i < 16; ++i) { if (begin + i < n) data[begin + i] = running; running += v[i]; }
  if (threadIdx.x == 255) sums[blockIdx.x] = s_total[255];
}
__global__ __launch_bounds__(256) void scan_sums(uint32_t* sums, uint32_t n_blocks) {
  __shared__ uint32_t s_total[256];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? sums[i] : 0u;
    s_total[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t step = 1; step < 256; step <<= 1) {
      const uint32_t add = threadIdx.x >= step ? s_total[threadIdx.x - step] : 0u;
      __syncthreads();
      s_total[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < n_blocks) sums[i] = s_carry + s_total[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry += s_total[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[n_blocks] = s_carry;
}
__global__ __launch_bounds__(256) void scan_add(uint32_t* data, uint64_t n, const uint32_t* sums) {
  const uint32_t offset = sums[blockIdx.x];
  const uint64_t begin = uint64_t{blockIdx.x} * SCAN_BLOCK;
  for (uint32_t i = threadIdx.x; i < SCAN_BLOCK && begin + i < n; i += 256) data[begin + i] += offset;
}

// Find or insert `tuple` in a workgroup's LDS table (tags / keys as in aggregate_rows, same lock discipline as global_slot).
// Returns the slot, or 0xFFFFFFFF when the table is full.
__device__ __forceinline__ uint32_t lds_slot(uint32_t* s_tags, uint64_t* s_keys, uint32_t slots, uint32_t words, const uint64_t (&tuple)[MAX_GROUPBY + 1], uint64_t hash,
                                             uint32_t* s_n_groups) {
  const uint32_t ready = 0x80000000u | static_cast<uint32_t>(hash >> 33);
  uint32_t slot = static_cast<uint32_t>(hash) & (slots - 1), probes = 0, result = 0xFFFFFFFFu;
  bool done = false;
  while (!done) {
    const uint32_t tag = __hip_atomic_load(&s_tags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_ACQUIRE);
    if (tag == TAG_EMPTY) {
      if (atomicCAS(&s_tags[slot], TAG_EMPTY, TAG_LOCKED) == TAG_EMPTY) {
#pragma unroll
        for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) { if (w < words) s_keys[slot * words + w] = tuple[w]; }
        atomicAdd(s_n_groups, 1u);
        __threadfence_block();
        atomicExch(&s_tags[slot], ready);
        result = slot;
        done = true;
      }
    } else if (tag != TAG_LOCKED) {
      bool equal = tag == ready;
      if (equal) {
#pragma unroll
        for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) { if (w < words) equal &= s_keys[slot * words + w] == tuple[w]; }
      }
      if (equal) {
        result = slot;
        done = true;
      } else {
        slot = (slot + 1) & (slots - 1);
        if (++probes >= slots) done = true;
      }
    }
  }
  return result;
}

// LDS layout: keys[S][words] u64 | first[S] u64 | last[S] u64 | values[S][A] u64 | counts[S][A] u32 | tags[S] u32 | groups, spilled u32
template <int WORDS, bool NARROW>
__global__ __launch_bounds__(256) void aggregate_partitions(AggArgs a, PartitionArgs p, DirectGroups direct) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t words = WORDS;
  const uint32_t slots = p.lds_slots, tid = threadIdx.x;
  uint64_t* s_keys = reinterpret_cast<uint64_t*>(smem);
  uint64_t* s_first = s_keys + size_t{slots} * words;
  uint64_t* s_last = s_first + slots;
  uint64_t* s_values = s_last + slots;
  uint32_t* s_counts = reinterpret_cast<uint32_t*>(s_values + size_t{slots} * a.n_aggregates);
  uint32_t* s_tags = s_counts + size_t{slots} * a.n_aggregates;
  uint32_t* s_n_groups = s_tags + slots;   // [0] groups in the table, [1] rows that did not fit, [2] the give-up flag as the workgroup saw it
  for (uint32_t s = tid; s < slots; s += 256) {
    s_tags[s] = TAG_EMPTY;
    s_first[s] = ~0ull;
    s_last[s] = 0;
    for (uint32_t g = 0; g < a.n_aggregates; ++g) {
      s_values[s * a.n_aggregates + g] = initial_value(a.aggregates[g].function);
      s_counts[s * a.n_aggregates + g] = 0;
    }
  }
  if (tid == 0) {
    s_n_groups[0] = s_n_groups[1] = 0;
    s_n_groups[2] = __hip_atomic_load(&a.overflow[FLAG_GIVE_UP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | __hip_atomic_load(&a.overflow[FLAG_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_n_groups[2]) return;
  const uint32_t partitions = 1u << p.bits, partition = blockIdx.x / p.split, share = blockIdx.x % p.split;
  const uint64_t partition_begin = p.offsets[size_t{partition} * p.n_parts];
  const uint64_t partition_end = partition + 1 < partitions ? p.offsets[size_t{partition + 1} * p.n_parts] : p.total_rows;
  const uint64_t per_share = (partition_end - partition_begin + p.split - 1) / p.split;
  const uint64_t begin = min(partition_end, partition_begin + share * per_share), end = min(partition_end, begin + per_share);
  const uint32_t pairs = p.record_words / 2;
#pragma unroll 1
  for (uint64_t base = begin; base < end; base += 256) {
    const uint64_t i = base + tid;
    if (i >= end) continue;
    uint64_t record[2 * MAX_RECORD_PAIRS];
    if (NARROW) {   // widened into the layout of the 64-bit records: head, tuple, the inputs' decoded words (contribution_from below)
      const u32x4* in = reinterpret_cast<const u32x4*>(p.records) + i * pairs;
      uint32_t narrow[4 * MAX_NARROW_QUADS];
#pragma unroll
      for (uint32_t k = 0; k < MAX_NARROW_QUADS; ++k) {
        u32x4 v; v.x = 0; v.y = 0; v.z = 0; v.w = 0;
        if (k < pairs) v = in[k];
        narrow[4 * k] = v.x; narrow[4 * k + 1] = v.y; narrow[4 * k + 2] = v.z; narrow[4 * k + 3] = v.w;
      }
#pragma unroll
      for (uint32_t k = 0; k < 2 * MAX_RECORD_PAIRS; ++k) record[k] = 0;
      record[0] = narrow[0] | static_cast<uint64_t>(narrow[1] & 0xFFFFu) << 32;
      record[1] = narrow[1] >> 16;
#pragma unroll
      for (int w = 1; w < WORDS; ++w) record[1 + w] = widen_word(narrow[1 + w], a.groupby[w - 1].is_float);
#pragma unroll
      for (uint32_t g = 0; g < MAX_AGGREGATES; ++g) {
        if (g < p.carried) record[1 + words + g] = contribution_from(a.aggregates[g], widen_word(narrow[1 + words + g], a.aggregates[g].is_float));
      }
    } else {
      const u64x2* in = reinterpret_cast<const u64x2*>(p.records) + i * pairs;
#pragma unroll
      for (uint32_t k = 0; k < MAX_RECORD_PAIRS; ++k) {
        u64x2 v; v.x = 0; v.y = 0;
        if (k < pairs) v = in[k];
        record[2 * k] = v.x;
        record[2 * k + 1] = v.y;
      }
    }
    const uint64_t head = record[0];
    const uint64_t global_row = head & 0xFFFFFFFFull;
    const uint32_t present = static_cast<uint32_t>(head >> 32);
    uint64_t tuple[MAX_GROUPBY + 1];
#pragma unroll
    for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) tuple[w] = w < words ? record[1 + w] : 0ull;
    const uint64_t hash = hash_tuple(tuple, words);
    // (the partition took the hash's top bits; both tables index with its low bits)
    const uint32_t slot = lds_slot(s_tags, s_keys, slots, words, tuple, hash, s_n_groups);
    uint32_t gslot = 0xFFFFFFFFu;
    if (slot == 0xFFFFFFFFu) {   // the partition has more groups than the table holds: this row goes to the global table directly
      atomicAdd(&s_n_groups[1], 1u);
      if (__hip_atomic_load(&a.overflow[FLAG_GIVE_UP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | __hip_atomic_load(&a.overflow[FLAG_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
      gslot = global_slot(a, tuple, words, hash);
      if (gslot == 0xFFFFFFFFu) { a.overflow[FLAG_OVERFLOW] = 1; continue; }
      atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(global_row));
      atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(global_row));
    } else {
      atomicMin(reinterpret_cast<unsigned long long*>(&s_first[slot]), static_cast<unsigned long long>(global_row));
      atomicMax(reinterpret_cast<unsigned long long*>(&s_last[slot]), static_cast<unsigned long long>(global_row));
    }
#pragma unroll
    for (uint32_t g = 0; g < MAX_AGGREGATES; ++g) {
      if (g >= a.n_aggregates) continue;
      const AggColumn& c = a.aggregates[g];
      uint64_t contribution = 0;
      if (g < p.carried) {
        contribution = record[1 + words + g];
        if (!((present >> g) & 1)) continue;   // NULL inputs leave the aggregate unchanged
      }
      if (slot == 0xFFFFFFFFu) { merge_global(a, gslot, g, contribution, 1); continue; }
      accumulate_lds(c, &s_values[slot * a.n_aggregates + g], contribution);
      atomicAdd(&s_counts[slot * a.n_aggregates + g], 1u);
    }
  }
  __syncthreads();
  if (tid == 0 && s_n_groups[1]) {
    const uint32_t before = atomicAdd(&a.overflow[FLAG_SPILLED], s_n_groups[1]);
    if (before + s_n_groups[1] > a.spill_limit) __hip_atomic_store(&a.overflow[FLAG_GIVE_UP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (direct.enabled) {
    __shared__ uint32_t s_direct[8];   // [0] first index of this workgroup's groups, [1] groups written so far, [2] refused, [4..7] per wave
    if (tid == 0) {
      const uint32_t n = s_n_groups[0];
      const uint32_t base = n ? atomicAdd(direct.counter, n) : 0u;
      s_direct[0] = base;
      s_direct[1] = 0;
      s_direct[2] = uint64_t{base} + n > direct.out_capacity;
      if (s_direct[2]) a.overflow[FLAG_OVERFLOW] = 1;
    }
    __syncthreads();
    if (s_direct[2]) return;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    for (uint32_t begin = 0; begin < slots; begin += 256) {
      const uint32_t s = begin + tid;
      const bool taken = s < slots && s_tags[s] != TAG_EMPTY;
      const uint64_t peers = __ballot(taken);
      if (lane == 0) s_direct[4 + wave] = static_cast<uint32_t>(__popcll(peers));
      __syncthreads();
      uint32_t before = s_direct[1];
      for (uint32_t w = 0; w < wave; ++w) before += s_direct[4 + w];
      const uint32_t step_total = s_direct[4] + s_direct[5] + s_direct[6] + s_direct[7];
      __syncthreads();
      if (tid == 0) s_direct[1] += step_total;
      if (!taken) continue;
      const size_t idx = size_t{s_direct[0]} + before + static_cast<uint32_t>(__popcll(peers & ((1ull << lane) - 1)));
      const bool stage = idx < direct.staged.capacity;
      for (uint32_t w = 0; w < words; ++w) {
        const uint64_t v = s_keys[s * words + w];
        direct.keys[idx * words + w] = v;
        if (stage) direct.staged.keys[idx * words + w] = v;
      }
      direct.first[idx] = s_first[s];
      direct.last[idx] = s_last[s];
      if (stage) { direct.staged.first[idx] = s_first[s]; direct.staged.last[idx] = s_last[s]; }
      for (uint32_t g = 0; g < a.n_aggregates; ++g) {
        const uint64_t v = s_values[s * a.n_aggregates + g], c = s_counts[s * a.n_aggregates + g];
        direct.values[idx * a.n_aggregates + g] = v;
        direct.counts[idx * a.n_aggregates + g] = c;
        if (stage) { direct.staged.values[idx * a.n_aggregates + g] = v; direct.staged.counts[idx * a.n_aggregates + g] = c; }
      }
    }
    return;
  }
  // the partition's groups: nobody else has them, every one is entered into the global table exactly once
  for (uint32_t s = tid; s < slots; s += 256) {
    if (s_tags[s] == TAG_EMPTY) continue;
    if (__hip_atomic_load(&a.overflow[FLAG_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    uint64_t tuple[MAX_GROUPBY + 1];
    for (uint32_t w = 0; w < words; ++w) tuple[w] = s_keys[s * words + w];
    const uint32_t gslot = global_slot(a, tuple, words, hash_tuple(tuple, words));
    if (gslot == 0xFFFFFFFFu) { a.overflow[FLAG_OVERFLOW] = 1; continue; }
    atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(s_first[s]));
    atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(s_last[s]));
    for (uint32_t g = 0; g < a.n_aggregates; ++g) merge_global(a, gslot, g, s_values[s * a.n_aggregates + g], s_counts[s * a.n_aggregates + g]);
  }
}

// ---- fused TableScan(s) -> Projection -> AggregateHash (hy_scan_project_aggregate) ----------------------------------------------
// One pass over a data table: a row is tested against every filter (the per-chunk jobs prepare_jobs normalised for hy_table_scan),
// the aggregates' input expressions are evaluated for the rows that pass, and the results go to the accumulators of the
// row's group -- a workgroup-private LDS table like aggregate_partitions', merged into the global table at the end of the
// chunk.  No PosList, no expression column: what the chain writes to HBM and reads back (8 bytes per surviving row and scan,
// 4-8 bytes per row and expression, twice) never exists.  One workgroup per CHUNK (its segment descriptors, jobs and input
// expressions are staged in LDS once, its groups reach the global table once), 8192 rows at a time, a quarter per wave:
//   scan      the filters, in plan order, over the wave's 2048 rows: a lane takes rows k * 64 + lane (k = 0..31), sixteen loads of a
//             filter in flight together (two memory round trips per filter and 2048 rows, not one per row batch).
//   compact   the surviving rows' numbers, in row order, into the wave's list in LDS (ballot + mbcnt: no atomics, no barrier).
//             A selective plan (TPC-H Q6: 2 % of the rows) does everything below on full waves of SURVIVORS.
//   rows      FUSED_ROWS list entries per lane and batch.  GROUP BY columns first -- all their loads issued before any is used
//             (decode_columns) -- tuples of VALUES (a chunk may hold any mix of encodings) into the workgroup's table;
//             rows whose group does not fit go to the global table directly.  Then the distinct columns the expressions read,
//             decoded ONCE per batch the same way; the expressions run on registers.
//   accumulate  the first FUSED_DENSE groups a chunk meets (TPC-H Q1 has four in the table, Q6 one) own FUSED_CELLS copies of every
//             accumulator and a lane adds to copy lane % FUSED_CELLS: one cell per group and aggregate serialises all 256 lanes of the
//             workgroup (measured: 5.8 ms of a 9.3 ms Q1 even with a wave reduction in front of every atomic).  Further groups use the
//             table's own cells (many groups: few conflicts).  The copies are folded into the table before it is merged.
// (the -D overrides are for A/B builds: tools/fused_variants.sh compiles them side by side and times them through HY_LIBRARY)
#ifndef HY_FUSED_LDS_SLOTS
#define HY_FUSED_LDS_SLOTS 128
#endif
#ifndef HY_FUSED_ROWS
#define HY_FUSED_ROWS 2
#endif
#ifndef HY_FUSED_WAVES
#define HY_FUSED_WAVES 4
#endif
#ifndef HY_FUSED_CELLS
#define HY_FUSED_CELLS 16
#endif
constexpr uint32_t FUSED_LDS_SLOTS = HY_FUSED_LDS_SLOTS;
constexpr int FUSED_ROWS = HY_FUSED_ROWS;
constexpr int FUSED_COLUMNS = 6;                    // distinct columns the aggregates' inputs may read
constexpr uint32_t FUSED_WAVE_ROWS = SLICE_ROWS / 4;   // 2048
constexpr uint32_t FUSED_VIEWS = HY_MAX_FILTERS + MAX_GROUPBY + FUSED_COLUMNS;
constexpr uint32_t FUSED_DENSE = 4;    // groups (the first the chunk meets) whose accumulators are spread over ...
constexpr uint32_t FUSED_CELLS = HY_FUSED_CELLS;   // ... this many copies each
constexpr int FUSED_WAVES = HY_FUSED_WAVES;        // waves per SIMD the register allocation aims at

struct FusedNode {
  uint64_t literal;             // HY_EXPR_LITERAL: the int64 value, the float's bits (low word) or the double's bits
  uint32_t kind;                // HY_EXPR_*
  uint32_t op;                  // HY_ARITH_*
  uint32_t type;                // result type of the node (HY_TYPE_NULL: the NULL literal)
  uint32_t column;              // HY_EXPR_COLUMN: index into FusedPlan::columns
};
struct FusedInput {             // input of one device accumulator, postfix
  FusedNode nodes[HY_MAX_EXPRESSION_NODES];
  uint32_t n_nodes;             // 0: COUNT(*)
  uint32_t type;
};
struct FusedFilter {
  const DevSegment* segments;
  const ScanJob* jobs;          // [n_chunks]
};
struct FusedPlan {
  FusedFilter filters[HY_MAX_FILTERS];
  const DevSegment* columns[FUSED_COLUMNS];
  FusedInput inputs[MAX_AGGREGATES];
  uint32_t n_filters;
  uint32_t n_columns;
  uint32_t lds_slots;           // of the workgroup's group table: FUSED_LDS_SLOTS, or a handful without GROUP BY (one group)
  uint32_t debug;               // HY_FUSED_DEBUG (timing experiments): 1 no accumulation, 2 no expressions, 4 no input columns, 8 no GROUP BY lookups, 16 scans only
};
enum : uint32_t { FLAG_PASSED = 4 };   // (two words: rows that passed the filters, the row count of the chain's aggregate input)

// What the kernel needs of a data segment, staged in LDS once per workgroup (a slice is one chunk): read through the descriptor
// tables in global memory these are vector loads of 48 bytes in front of every batch of data loads -- and twelve registers each.
struct ColumnView {
  const void* data;
  const void* aux;
  const uint64_t* nulls;
  uint32_t aux_size;
  uint8_t encoding, data_type, width, present;
};
__device__ __forceinline__ ColumnView view_of(const DevSegment& s) {
  return ColumnView{s.data, s.aux, s.nulls, s.aux_size, s.encoding, s.data_type, s.width, 1};
}
// a value every lane holds alike, moved to a scalar register: branches on it are scalar branches
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
template <typename T>
__device__ __forceinline__ const T* uniform(const T* p) {
  const uint64_t bits = reinterpret_cast<uint64_t>(p);
  return reinterpret_cast<const T*>(static_cast<uint64_t>(uniform(static_cast<uint32_t>(bits >> 32))) << 32 | uniform(static_cast<uint32_t>(bits)));
}
__device__ __forceinline__ ColumnView uniform(const ColumnView& v) {
  const uint32_t shape = uniform(static_cast<uint32_t>(v.encoding) | static_cast<uint32_t>(v.data_type) << 8 | static_cast<uint32_t>(v.width) << 16 | static_cast<uint32_t>(v.present) << 24);
  return ColumnView{uniform(v.data), uniform(v.aux), uniform(v.nulls), uniform(v.aux_size), static_cast<uint8_t>(shape), static_cast<uint8_t>(shape >> 8), static_cast<uint8_t>(shape >> 16),
                    static_cast<uint8_t>(shape >> 24)};
}
__device__ __forceinline__ ScanJob uniform(const ScanJob& j) {
  ScanJob u;
  u.mode = uniform(j.mode); u.kind = uniform(j.kind); u.flags = uniform(j.flags); u.null_vid = uniform(j.null_vid);
  u.lo = static_cast<uint64_t>(uniform(static_cast<uint32_t>(j.lo >> 32))) << 32 | uniform(static_cast<uint32_t>(j.lo));
  u.span = static_cast<uint64_t>(uniform(static_cast<uint32_t>(j.span >> 32))) << 32 | uniform(static_cast<uint32_t>(j.span));
  return u;
}

// One row of a data segment against its chunk's job (what eval_row of scan.hip decides for JOB_SCAN jobs of data segments).
__device__ __forceinline__ bool filter_row(const ColumnView& s, const ScanJob& job, uint32_t row) {
  if (s.encoding == HY_ENC_MVCC) {   // a Validate filter: Validate::is_row_visible (validate.cpp:47-55) on the chunk's tids / begin cids / end cids
    const uint32_t snapshot = static_cast<uint32_t>(job.lo), our_tid = static_cast<uint32_t>(job.span);
    const uint32_t tid = static_cast<const uint32_t*>(s.data)[row], begin = static_cast<const uint32_t*>(s.aux)[row], end = reinterpret_cast<const uint32_t*>(s.nulls)[row];
    return snapshot < end && ((snapshot >= begin) != (tid == our_tid));
  }
  const bool invert = job.flags & JF_INVERT;
  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = aload_compressed(s.data, s.width, row);
    if (job.kind == KIND_VALUE_ID_SET) return vid < job.null_vid && ((reinterpret_cast<const uint64_t*>(job.lo)[vid >> 6] >> (vid & 63)) & 1) != 0;
    const bool in = (vid - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
    return (in != invert) && vid != job.null_vid;   // (the NULL test itself: null_vid = 0xFFFFFFFF)
  }
  const bool is_null = s.nulls ? ((s.nulls[row >> 6] >> (row & 63)) & 1) != 0 : false;
  if (job.kind == KIND_NULLTEST) return is_null != invert;
  if (is_null) return false;   // NULL never matches a comparison
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

// The filter over a wave's part of the slice: bit k = row first_row + k * 64 + lane matches (rows at or behind n_rows: undefined, the
// caller masks them).  Stored words of at most four bytes -- value ids, FrameOfReference offsets, int / float values -- are loaded
// sixteen rows at a time before the first is tested; 8-byte values and value-id sets go row by row.
__device__ __forceinline__ uint32_t filter_wave_rows(const ColumnView& s, const ScanJob& job, uint32_t first_row, uint32_t lane, uint32_t n_rows) {
  constexpr int K = static_cast<int>(FUSED_WAVE_ROWS / 64), B = 16;
  const bool invert = job.flags & JF_INVERT;
  const bool words = s.encoding != HY_ENC_MVCC && (s.encoding == HY_ENC_DICTIONARY || s.encoding == HY_ENC_FRAME_OF_REFERENCE || job.kind == KIND_U32 || job.kind == KIND_F32 || job.kind == KIND_NULLTEST);
  uint32_t bits = 0;
  if (!words || job.kind == KIND_VALUE_ID_SET) {
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const uint32_t r = static_cast<uint32_t>(k) * 64 + lane;
      if (r < n_rows && filter_row(s, job, first_row + r)) bits |= 1u << k;
    }
    return bits;
  }
  const uint32_t width = (s.encoding == HY_ENC_DICTIONARY || s.encoding == HY_ENC_FRAME_OF_REFERENCE) ? s.width : 4u;
  const uint32_t lo = static_cast<uint32_t>(job.lo), span = static_cast<uint32_t>(job.span), null_vid = job.null_vid;
  const float lower = __uint_as_float(lo), upper = __uint_as_float(span);
  const bool lower_inclusive = job.flags & JF_LOWER_INCL, upper_inclusive = job.flags & JF_UPPER_INCL;
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += B) {
    if (static_cast<uint32_t>(k0) * 64 >= n_rows) break;
    uint32_t row[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const uint32_t r = static_cast<uint32_t>(k0 + k) * 64 + lane;
      row[k] = first_row + (r < n_rows ? r : 0);
    }
    uint32_t null_bits = 0, round = 0;
    if (s.encoding != HY_ENC_DICTIONARY && s.nulls) {
      uint64_t word[B];
#pragma unroll
      for (int k = 0; k < B; ++k) word[k] = s.nulls[row[k] >> 6];
#pragma unroll
      for (int k = 0; k < B; ++k) null_bits |= static_cast<uint32_t>((word[k] >> (row[k] & 63)) & 1) << k;
    }
    if (job.kind == KIND_NULLTEST && s.encoding != HY_ENC_DICTIONARY) {
      bits |= ((invert ? ~null_bits : null_bits) & 0xFFFFu) << k0;
      continue;
    }
    uint32_t raw[B];
    if (width == 1) {
#pragma unroll
      for (int k = 0; k < B; ++k) raw[k] = static_cast<const uint8_t*>(s.data)[row[k]];
    } else if (width == 2) {
#pragma unroll
      for (int k = 0; k < B; ++k) raw[k] = static_cast<const uint16_t*>(s.data)[row[k]];
    } else {
#pragma unroll
      for (int k = 0; k < B; ++k) raw[k] = static_cast<const uint32_t*>(s.data)[row[k]];
    }
    if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
      uint32_t bias[B];
#pragma unroll
      for (int k = 0; k < B; ++k) bias[k] = static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row[k] / HY_FOR_BLOCK_SIZE]);
#pragma unroll
      for (int k = 0; k < B; ++k) raw[k] += bias[k];
    }
    if (s.encoding == HY_ENC_DICTIONARY) {
#pragma unroll
      for (int k = 0; k < B; ++k) round |= ((((raw[k] - lo) <= span) != invert) && raw[k] != null_vid ? 1u : 0u) << k;
    } else {
      if (job.kind == KIND_F32) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
          const float x = __uint_as_float(raw[k]);
          round |= (((lower_inclusive ? x >= lower : x > lower) && (upper_inclusive ? x <= upper : x < upper)) ? 1u : 0u) << k;
        }
      } else {
#pragma unroll
        for (int k = 0; k < B; ++k) round |= ((raw[k] - lo) <= span ? 1u : 0u) << k;
      }
      round = (invert ? ~round : round) & ~null_bits;
    }
    bits |= (round & 0xFFFFu) << k0;
  }
  return bits;
}

// What the stages of decode_columns must know of the staged views without loading them: bit v = view v ...
struct ViewMasks {
  uint32_t present;      // ... exists
  uint32_t nullable;     // ... is a value / FrameOfReference segment with a null bitmap
  uint32_t dictionary;   // ... is a dictionary segment
  uint32_t frame;        // ... is a FrameOfReference segment
  uint32_t is_int;       // ... holds int32 values
  uint32_t is_float;     // ... holds float values
};

// N columns (views first .. first + N - 1) for FUSED_ROWS rows of a lane, stage by stage, so that the loads of ALL columns of a stage are in
// flight together: the stored words (value ids / offsets / values), the null words of value segments, the dictionary entries.  Values
// come out as int64 (int / long columns) or as the bits of the double; float columns as the bits of the double too (WIDEN: GROUP BY
// keys, like decode_rows) or as the float's own bits in the low word (the expressions compute on those).  Data segments only.
template <int N, bool WIDEN>
__device__ __forceinline__ void decode_columns(const ColumnView* views, const ViewMasks& all, uint32_t first, const uint32_t (&row)[FUSED_ROWS], uint64_t (&value)[N][FUSED_ROWS],
                                               uint32_t (&nulls)[N]) {
  constexpr int R = FUSED_ROWS;
  const uint32_t present = all.present >> first, nullable = all.nullable >> first, dictionary = all.dictionary >> first, frame = all.frame >> first, is_int = all.is_int >> first,
                 is_float = all.is_float >> first;
#pragma unroll
  for (int c = 0; c < N; ++c) {   // stage 1: stored words
    nulls[c] = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) value[c][i] = 0;
    if (!((present >> c) & 1)) continue;
    const ColumnView s = uniform(views[first + c]);
    if (s.encoding == HY_ENC_DICTIONARY || s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
      if (s.width == 1) {
#pragma unroll
        for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint8_t*>(s.data)[row[i]];
      } else if (s.width == 2) {
#pragma unroll
        for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint16_t*>(s.data)[row[i]];
      } else {
#pragma unroll
        for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint32_t*>(s.data)[row[i]];
      }
      if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {   // the block minimum rides in the upper half
#pragma unroll
        for (int i = 0; i < R; ++i) value[c][i] |= static_cast<uint64_t>(static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row[i] / HY_FOR_BLOCK_SIZE])) << 32;
      }
    } else if (s.data_type == HY_TYPE_INT || s.data_type == HY_TYPE_FLOAT) {
#pragma unroll
      for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint32_t*>(s.data)[row[i]];
    } else {
#pragma unroll
      for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint64_t*>(s.data)[row[i]];
    }
  }
#pragma unroll
  for (int c = 0; c < N; ++c) {   // stage 1b: NULLs of value / FrameOfReference segments
    if (!((nullable >> c) & 1)) continue;
    const uint64_t* null_words = uniform(views[first + c].nulls);
    uint64_t word[R];
#pragma unroll
    for (int i = 0; i < R; ++i) word[i] = null_words[row[i] >> 6];
#pragma unroll
    for (int i = 0; i < R; ++i) nulls[c] |= static_cast<uint32_t>((word[i] >> (row[i] & 63)) & 1) << i;
  }
#pragma unroll
  for (int c = 0; c < N; ++c) {   // stage 2: dictionary entries
    if (!((dictionary >> c) & 1)) continue;
    const ColumnView s = uniform(views[first + c]);
    uint32_t index[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      index[i] = static_cast<uint32_t>(value[c][i]);
      if (index[i] >= s.aux_size) { nulls[c] |= 1u << i; index[i] = 0; }   // the NULL value id
    }
    if (s.aux_size == 0) {
#pragma unroll
      for (int i = 0; i < R; ++i) value[c][i] = 0;
    } else if (s.data_type == HY_TYPE_INT || s.data_type == HY_TYPE_FLOAT) {
#pragma unroll
      for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint32_t*>(s.aux)[index[i]];
    } else {
#pragma unroll
      for (int i = 0; i < R; ++i) value[c][i] = static_cast<const uint64_t*>(s.aux)[index[i]];
    }
  }
#pragma unroll
  for (int c = 0; c < N; ++c) {   // stage 3: stored word -> value
    if (!((present >> c) & 1)) continue;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint64_t raw = value[c][i];
      if ((frame >> c) & 1) value[c][i] = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(raw) + static_cast<uint32_t>(raw >> 32))));
      else if ((is_int >> c) & 1) value[c][i] = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(raw))));
      else if (WIDEN && ((is_float >> c) & 1)) value[c][i] = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(__uint_as_float(static_cast<uint32_t>(raw)))));
    }
  }
}

// A value on the expression stack (int64 | float bits in the low word | double bits) <-> the Value arithmetic_cell computes on
__device__ __forceinline__ Value stack_value(uint64_t bits, uint32_t type) {
  Value v{false, 0, 0.0};
  if (type == HY_TYPE_FLOAT) v.f = static_cast<double>(__uint_as_float(static_cast<uint32_t>(bits)));
  else if (type == HY_TYPE_DOUBLE) v.f = __longlong_as_double(static_cast<long long>(bits));
  else v.i = static_cast<int64_t>(bits);
  return v;
}
__device__ __forceinline__ uint64_t stack_bits(const Value& v, uint32_t type) {
  if (type == HY_TYPE_FLOAT) return __float_as_uint(static_cast<float>(v.f));
  if (type == HY_TYPE_DOUBLE) return static_cast<uint64_t>(__double_as_longlong(v.f));
  return static_cast<uint64_t>(v.i);
}

// The input expression of one accumulator for FUSED_ROWS rows of a lane: a three-slot stack held in registers (slot 0 = top; push and
// pop move the slots, nothing is indexed); column nodes take the batch's decoded columns.  + - * of two operands of the result's own
// type (the host gives literals the type the operation converts them to anyway) are single IEEE / integer operations on the stack
// words; everything else goes through arithmetic_cell -- the cell hy_projection_arithmetic computes.  Floats leave as double bits.
__device__ __forceinline__ void evaluate_input(const FusedInput& input, const uint64_t (&columns)[FUSED_COLUMNS][FUSED_ROWS], const uint32_t (&column_nulls)[FUSED_COLUMNS], uint32_t valid,
                                               uint64_t (&out)[FUSED_ROWS], uint32_t* out_nulls) {
  uint64_t s0[FUSED_ROWS], s1[FUSED_ROWS], s2[FUSED_ROWS];
  uint32_t n0 = 0, n1 = 0, n2 = 0, t0 = HY_TYPE_NULL, t1 = HY_TYPE_NULL, t2 = HY_TYPE_NULL;
#pragma unroll
  for (int i = 0; i < FUSED_ROWS; ++i) s0[i] = s1[i] = s2[i] = 0;
  const uint32_t n_nodes = uniform(input.n_nodes);
#pragma unroll 1
  for (uint32_t k = 0; k < n_nodes; ++k) {
    const FusedNode& node = input.nodes[k];
    const uint32_t kind = uniform(node.kind), type = uniform(node.type);
    if (kind == HY_EXPR_ARITHMETIC) {   // slot 1 <op> slot 0 -> slot 0; slot 2 moves up
      const uint32_t op = uniform(node.op);
      uint32_t nulls = n0 | n1;
      if (op <= HY_ARITH_MUL && t0 == type && t1 == type) {
        if (type == HY_TYPE_FLOAT) {
#pragma unroll
          for (int i = 0; i < FUSED_ROWS; ++i) {
            const float x = __uint_as_float(static_cast<uint32_t>(s1[i])), y = __uint_as_float(static_cast<uint32_t>(s0[i]));
            s0[i] = __float_as_uint(op == HY_ARITH_ADD ? __fadd_rn(x, y) : op == HY_ARITH_SUB ? __fsub_rn(x, y) : __fmul_rn(x, y));
          }
        } else if (type == HY_TYPE_DOUBLE) {
#pragma unroll
          for (int i = 0; i < FUSED_ROWS; ++i) {
            const double x = __longlong_as_double(static_cast<long long>(s1[i])), y = __longlong_as_double(static_cast<long long>(s0[i]));
            s0[i] = static_cast<uint64_t>(__double_as_longlong(op == HY_ARITH_ADD ? __dadd_rn(x, y) : op == HY_ARITH_SUB ? __dsub_rn(x, y) : __dmul_rn(x, y)));
          }
        } else if (type == HY_TYPE_LONG) {
#pragma unroll
          for (int i = 0; i < FUSED_ROWS; ++i) s0[i] = op == HY_ARITH_ADD ? s1[i] + s0[i] : op == HY_ARITH_SUB ? s1[i] - s0[i] : s1[i] * s0[i];
        } else {
#pragma unroll
          for (int i = 0; i < FUSED_ROWS; ++i) {
            const uint32_t x = static_cast<uint32_t>(s1[i]), y = static_cast<uint32_t>(s0[i]);
            s0[i] = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(op == HY_ARITH_ADD ? x + y : op == HY_ARITH_SUB ? x - y : x * y)));
          }
        }
      } else if (op <= HY_ARITH_MUL) {   // + - * with conversions: the rows side by side, the operator a constant of each instance
#pragma unroll
        for (int i = 0; i < FUSED_ROWS; ++i) {
          Value result{false, 0, 0.0};
          if (((valid & ~nulls) >> i) & 1) {
            const Value x = stack_value(s1[i], t1), y = stack_value(s0[i], t0);
            if (op == HY_ARITH_ADD) arithmetic_cell(HY_ARITH_ADD, t1, t0, type, x, y, &result);
            else if (op == HY_ARITH_SUB) arithmetic_cell(HY_ARITH_SUB, t1, t0, type, x, y, &result);
            else arithmetic_cell(HY_ARITH_MUL, t1, t0, type, x, y, &result);
          }
          s0[i] = stack_bits(result, type);
        }
      } else {   // / and %: 64-bit division, fmod -- long instruction sequences with many live registers: ONE instance, the rows one after the other
                 // (row i selected by comparisons: an index would put the stack into scratch memory)
#pragma unroll 1
        for (int i = 0; i < FUSED_ROWS; ++i) {
          uint64_t x = s1[0], y = s0[0];
#pragma unroll
          for (int j = 1; j < FUSED_ROWS; ++j) { x = i == j ? s1[j] : x; y = i == j ? s0[j] : y; }
          Value result{false, 0, 0.0};
          bool is_null = true;
          if (((valid & ~nulls) >> i) & 1) is_null = arithmetic_cell(op, t1, t0, type, stack_value(x, t1), stack_value(y, t0), &result);   // NULL: division / modulo by zero
          if (is_null) nulls |= (valid & (1u << i));
          const uint64_t bits = stack_bits(result, type);
#pragma unroll
          for (int j = 0; j < FUSED_ROWS; ++j) s0[j] = i == j ? bits : s0[j];
        }
      }
#pragma unroll
      for (int i = 0; i < FUSED_ROWS; ++i) s1[i] = s2[i];
      n0 = nulls; t0 = type;
      n1 = n2; t1 = t2;
    } else {                            // push
#pragma unroll
      for (int i = 0; i < FUSED_ROWS; ++i) { s2[i] = s1[i]; s1[i] = s0[i]; }
      n2 = n1; t2 = t1;
      n1 = n0; t1 = t0;
      t0 = type;
      if (kind == HY_EXPR_COLUMN) {
        const uint32_t which = uniform(node.column);
#pragma unroll
        for (int c = 0; c < FUSED_COLUMNS; ++c) {
          if (which != static_cast<uint32_t>(c)) continue;
#pragma unroll
          for (int i = 0; i < FUSED_ROWS; ++i) s0[i] = columns[c][i];
          n0 = column_nulls[c];
        }
      } else {
        const uint64_t literal = node.literal;
#pragma unroll
        for (int i = 0; i < FUSED_ROWS; ++i) s0[i] = literal;
        n0 = type == HY_TYPE_NULL ? 0xFFFFFFFFu : 0u;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < FUSED_ROWS; ++i) out[i] = t0 == HY_TYPE_FLOAT ? static_cast<uint64_t>(__double_as_longlong(static_cast<double>(__uint_as_float(static_cast<uint32_t>(s0[i]))))) : s0[i];
  *out_nulls = n0;
}

// Find or insert `tuple` in the workgroup's table (lds_slot with one addition: a new group gets its arrival number -- its dense index).
__device__ __forceinline__ uint32_t fused_lds_slot(uint32_t* s_tags, uint64_t* s_keys, uint32_t* s_dense_of_slot, uint32_t* s_slot_of_dense, uint32_t slots, uint32_t words,
                                                   const uint64_t (&tuple)[MAX_GROUPBY + 1], uint64_t hash, uint32_t* s_n_groups) {
  const uint32_t ready = 0x80000000u | static_cast<uint32_t>(hash >> 33);
  uint32_t slot = static_cast<uint32_t>(hash) & (slots - 1), probes = 0, result = 0xFFFFFFFFu;
  bool done = false;
  while (!done) {
    const uint32_t tag = __hip_atomic_load(&s_tags[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_ACQUIRE);
    if (tag == TAG_EMPTY) {
      if (atomicCAS(&s_tags[slot], TAG_EMPTY, TAG_LOCKED) == TAG_EMPTY) {
#pragma unroll
        for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) { if (w < words) s_keys[slot * words + w] = tuple[w]; }
        const uint32_t arrival = atomicAdd(s_n_groups, 1u);
        s_dense_of_slot[slot] = arrival;
        if (arrival < FUSED_DENSE) s_slot_of_dense[arrival] = slot;
        __threadfence_block();
        atomicExch(&s_tags[slot], ready);
        result = slot;
        done = true;
      }
    } else if (tag != TAG_LOCKED) {
      bool equal = tag == ready;
      if (equal) {
#pragma unroll
        for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) { if (w < words) equal &= s_keys[slot * words + w] == tuple[w]; }
      }
      if (equal) {
        result = slot;
        done = true;
      } else {
        slot = (slot + 1) & (slots - 1);
        if (++probes >= slots) done = true;
      }
    }
  }
  return result;
}

// LDS layout: keys[S][words] u64 | first[S] u64 | last[S] u64 | values[S][A] u64 | counts[S][A] u32 | tags[S] u32 | dense index of a slot [S] u32 |
//             groups, passed, stop, -, slot of a dense index [4], view masks [6] u32 (+ pad to 64 bytes) | the waves' lists of surviving rows [4][2048] u16 |
//             this chunk's segment views [FUSED_VIEWS] | the filters' jobs [HY_MAX_FILTERS] | the input expressions [A] |
//             the dense groups' cells: values [4][A + 2][16] u64 (the last two: first / last row) | counts [4][A][16] u32
__global__ __launch_bounds__(256, FUSED_WAVES) void fused_rows(AggArgs a, const FusedPlan* __restrict__ plan, uint32_t n_chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = FUSED_ROWS;
  const uint32_t slots = plan->lds_slots;
  const uint32_t words = a.n_groupby + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_aggregates = a.n_aggregates;
  uint64_t* s_keys = reinterpret_cast<uint64_t*>(smem);
  uint64_t* s_first = s_keys + size_t{slots} * words;
  uint64_t* s_last = s_first + slots;
  uint64_t* s_values = s_last + slots;
  uint32_t* s_counts = reinterpret_cast<uint32_t*>(s_values + size_t{slots} * n_aggregates);
  uint32_t* s_tags = s_counts + size_t{slots} * n_aggregates;
  uint32_t* s_dense_of_slot = s_tags + slots;
  uint32_t* s_n_groups = s_dense_of_slot + slots;   // [0] groups in the table, [1] rows that passed the filters, [2] the overflow flag as the workgroup saw it
  uint32_t* s_slot_of_dense = s_n_groups + 4;       // [FUSED_DENSE]
  uint16_t* s_list = reinterpret_cast<uint16_t*>(s_n_groups + 16) + size_t{wave} * FUSED_WAVE_ROWS;
  ColumnView* s_views = reinterpret_cast<ColumnView*>(reinterpret_cast<uint16_t*>(s_n_groups + 16) + SLICE_ROWS);   // filters [0, 4) | GROUP BY [4, 8) | inputs' columns [8, 8 + C)
  ScanJob* s_jobs = reinterpret_cast<ScanJob*>(s_views + FUSED_VIEWS);
  ViewMasks* s_masks = reinterpret_cast<ViewMasks*>(s_n_groups + 8);
  FusedInput* s_inputs = reinterpret_cast<FusedInput*>(s_jobs + HY_MAX_FILTERS);   // the accumulators' input expressions (read node by node for every batch: global memory is a round trip each)
  // The cells of the first FUSED_DENSE groups the chunk meets: FUSED_CELLS copies of every accumulator, a lane adds to copy lane % FUSED_CELLS.  256 lanes
  // adding to the four groups of TPC-H Q1 through ONE cell per group and aggregate serialise completely; sixteen copies take a sixteenth of that.
  uint64_t* s_cell_values = reinterpret_cast<uint64_t*>(s_inputs + n_aggregates);                  // [FUSED_DENSE][A + 2][FUSED_CELLS]
  uint32_t* s_cell_counts = reinterpret_cast<uint32_t*>(s_cell_values + FUSED_DENSE * (n_aggregates + 2) * FUSED_CELLS);   // [FUSED_DENSE][A][FUSED_CELLS]
  const uint32_t chunk = blockIdx.x;
  for (uint32_t s = tid; s < slots; s += 256) {
    s_tags[s] = TAG_EMPTY;
    s_first[s] = ~0ull;
    s_last[s] = 0;
    for (uint32_t g = 0; g < n_aggregates; ++g) {
      s_values[s * n_aggregates + g] = initial_value(a.aggregates[g].function);
      s_counts[s * n_aggregates + g] = 0;
    }
  }
  for (uint32_t cell = tid; cell < FUSED_DENSE * (n_aggregates + 2) * FUSED_CELLS; cell += 256) {
    const uint32_t g = (cell / FUSED_CELLS) % (n_aggregates + 2);
    s_cell_values[cell] = g < n_aggregates ? initial_value(a.aggregates[g].function) : (g == n_aggregates ? ~0ull : 0ull);
  }
  for (uint32_t cell = tid; cell < FUSED_DENSE * n_aggregates * FUSED_CELLS; cell += 256) s_cell_counts[cell] = 0;
  if (tid == 0) {
    s_n_groups[0] = s_n_groups[1] = 0;
    s_n_groups[2] = __hip_atomic_load(&a.overflow[FLAG_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {   // this chunk's descriptors and jobs, the input expressions -> LDS
    const uint32_t n_filters = plan->n_filters, n_columns = plan->n_columns;
    const DevSegment* segments = nullptr;
#pragma unroll
    for (uint32_t f = 0; f < HY_MAX_FILTERS; ++f) if (tid == f && f < n_filters) segments = plan->filters[f].segments;
#pragma unroll
    for (uint32_t g = 0; g < MAX_GROUPBY; ++g) if (tid == HY_MAX_FILTERS + g && g < a.n_groupby) segments = a.groupby[g].segments;
#pragma unroll
    for (uint32_t c = 0; c < static_cast<uint32_t>(FUSED_COLUMNS); ++c) if (tid == HY_MAX_FILTERS + MAX_GROUPBY + c && c < n_columns) segments = plan->columns[c];
    const ColumnView view = segments ? view_of(segments[chunk]) : ColumnView{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
    if (tid < FUSED_VIEWS) s_views[tid] = view;
    if (wave == 0) {   // (the views sit in lanes 0 .. FUSED_VIEWS - 1 of wave 0: a ballot per property)
      const bool mine = tid < FUSED_VIEWS && view.present;
      ViewMasks masks;
      masks.present = static_cast<uint32_t>(__ballot(mine));
      masks.nullable = static_cast<uint32_t>(__ballot(mine && view.encoding != HY_ENC_DICTIONARY && view.nulls != nullptr));
      masks.dictionary = static_cast<uint32_t>(__ballot(mine && view.encoding == HY_ENC_DICTIONARY));
      masks.frame = static_cast<uint32_t>(__ballot(mine && view.encoding == HY_ENC_FRAME_OF_REFERENCE));
      masks.is_int = static_cast<uint32_t>(__ballot(mine && view.data_type == HY_TYPE_INT));
      masks.is_float = static_cast<uint32_t>(__ballot(mine && view.data_type == HY_TYPE_FLOAT));
      if (tid == 0) *s_masks = masks;
    }
    if (tid >= 64 && tid < 64 + n_filters) s_jobs[tid - 64] = plan->filters[tid - 64].jobs[chunk];
    const uint64_t* plan_inputs = reinterpret_cast<const uint64_t*>(plan->inputs);
    uint64_t* staged_inputs = reinterpret_cast<uint64_t*>(s_inputs);
    for (uint32_t w = tid; w < n_aggregates * (sizeof(FusedInput) / 8); w += 256) staged_inputs[w] = plan_inputs[w];
  }
  __syncthreads();
  if (s_n_groups[2]) return;   // the global table is too small: the host retries with a larger one
  const uint64_t chunk_base = a.row_base[chunk];
  const uint32_t chunk_rows = static_cast<uint32_t>(a.row_base[chunk + 1] - chunk_base);
  const uint32_t debug = plan->debug;
  const uint32_t n_filters = plan->n_filters;
  uint32_t wave_passed = 0;
  ViewMasks masks;
  masks.present = uniform(s_masks->present); masks.nullable = uniform(s_masks->nullable); masks.dictionary = uniform(s_masks->dictionary);
  masks.frame = uniform(s_masks->frame); masks.is_int = uniform(s_masks->is_int); masks.is_float = uniform(s_masks->is_float);

  // The chunk, 8192 rows at a time; every wave takes a quarter of the piece (no barrier in here: the waves share the group table through
  // LDS atomics only).
#pragma unroll 1
  for (uint32_t piece = 0; piece < chunk_rows; piece += SLICE_ROWS) {
    const uint32_t first_row = piece + wave * FUSED_WAVE_ROWS;   // chunk offset of the wave's first row
    if (first_row >= chunk_rows) break;
    const uint32_t wave_rows = min(FUSED_WAVE_ROWS, chunk_rows - first_row);
    const uint64_t wave_base = chunk_base + first_row;           // its global row number
    uint32_t n_list = 0;
    {
      // ---- the scans: every filter over the wave's rows ---------------------------------------------------------------------------
      uint32_t pass = 0;
#pragma unroll
      for (uint32_t k = 0; k < FUSED_WAVE_ROWS / 64; ++k) pass |= (k * 64 + lane < wave_rows ? 1u : 0u) << k;
#pragma unroll 1
      for (uint32_t f = 0; f < n_filters; ++f) {
        const ScanJob job = uniform(s_jobs[f]);
        if (job.mode == JOB_ALL) continue;                                       // the chunk's early-out: every row matches
        if (job.mode == JOB_NONE || (job.flags & JF_NEVER)) { pass = 0; break; }   // ... or none does
        pass &= filter_wave_rows(uniform(s_views[f]), job, first_row, lane, wave_rows);
        if (__ballot(pass != 0) == 0) break;
      }
      // ---- the survivors' row numbers (relative to the quarter), in row order -----------------------------------------------------
#pragma unroll
      for (uint32_t k = 0; k < FUSED_WAVE_ROWS / 64; ++k) {
        const bool alive = (pass >> k) & 1;
        const uint64_t lanes = __ballot(alive);
        if (alive) s_list[n_list + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(lanes >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(lanes), 0u))] = static_cast<uint16_t>(k * 64 + lane);
        n_list += static_cast<uint32_t>(__popcll(lanes));
      }
      wave_passed += n_list;
    }
    if (debug & 16) n_list = 0;

#pragma unroll 1
    for (uint32_t base = 0; base < n_list; base += 64 * R) {
      uint32_t row[R], offset[R], pass = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const uint32_t entry = base + static_cast<uint32_t>(i) * 64 + lane;
        if (entry < n_list) pass |= 1u << i;
        offset[i] = s_list[entry < n_list ? entry : base];
        row[i] = first_row + offset[i];
      }
      // ---- GROUP BY tuples and their slots; `cell`: where the row's accumulators are ------------------------------------------------------
      uint32_t slot[R], dense[R], in_global = 0;
      {
        uint64_t key[MAX_GROUPBY][R];
        uint32_t key_nulls[MAX_GROUPBY];
        decode_columns<static_cast<int>(MAX_GROUPBY), true>(s_views, masks, HY_MAX_FILTERS, row, key, key_nulls);
        if (debug & 8) pass = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          slot[i] = 0xFFFFFFFFu;
          dense[i] = 0xFFFFFFFFu;
          uint64_t tuple[MAX_GROUPBY + 1];
          tuple[0] = 0;
#pragma unroll
          for (uint32_t g = 0; g < MAX_GROUPBY; ++g) {
            tuple[g + 1] = 0;
            if (g >= a.n_groupby) continue;
            if ((key_nulls[g] >> i) & 1) { tuple[0] |= 1ull << g; continue; }
            uint64_t bits = key[g][i];
            if (a.groupby[g].is_float && __longlong_as_double(static_cast<long long>(bits)) == 0.0) bits = 0;   // -0.0 and 0.0 are one group
            tuple[g + 1] = bits;
          }
          if (!((pass >> i) & 1)) continue;
          const uint64_t global_row = wave_base + offset[i];
          const uint64_t hash = hash_tuple_in_registers(tuple, words);
          slot[i] = fused_lds_slot(s_tags, s_keys, s_dense_of_slot, s_slot_of_dense, slots, words, tuple, hash, s_n_groups);
          if (slot[i] != 0xFFFFFFFFu) {
            dense[i] = s_dense_of_slot[slot[i]];
            // first / last row of the group: the dense groups' cells, or the slot itself
            const bool is_dense = dense[i] < FUSED_DENSE;
            uint64_t* first = is_dense ? &s_cell_values[(dense[i] * (n_aggregates + 2) + n_aggregates) * FUSED_CELLS + (lane % FUSED_CELLS)] : &s_first[slot[i]];
            uint64_t* last = is_dense ? first + FUSED_CELLS : &s_last[slot[i]];
            atomicMin(reinterpret_cast<unsigned long long*>(first), static_cast<unsigned long long>(global_row));
            atomicMax(reinterpret_cast<unsigned long long*>(last), static_cast<unsigned long long>(global_row));
            continue;
          }
          // the chunk has more groups than the table holds: this row goes to the global table directly
          uint64_t spilled[MAX_GROUPBY + 1];   // (global_slot walks its tuple in memory: a copy made on this path only keeps `tuple` in registers)
#pragma unroll
          for (uint32_t w = 0; w <= MAX_GROUPBY; ++w) spilled[w] = tuple[w];
          const uint32_t gslot = __hip_atomic_load(&a.overflow[FLAG_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0xFFFFFFFFu : global_slot(a, spilled, words, hash);
          if (gslot == 0xFFFFFFFFu) {
            a.overflow[FLAG_OVERFLOW] = 1;
            pass &= ~(1u << i);   // (the whole pass is repeated with a larger table)
            continue;
          }
          atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(global_row));
          atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(global_row));
          slot[i] = gslot;
          in_global |= 1u << i;
        }
      }

      // ---- the columns the expressions read, once per batch; then the aggregates: input expression, accumulators ------------------------
      uint64_t column[FUSED_COLUMNS][R];
      uint32_t column_nulls[FUSED_COLUMNS];
      if (!(debug & 4)) decode_columns<FUSED_COLUMNS, false>(s_views, masks, HY_MAX_FILTERS + MAX_GROUPBY, row, column, column_nulls);
      else {
#pragma unroll
        for (int c = 0; c < FUSED_COLUMNS; ++c) { column_nulls[c] = 0; for (int i = 0; i < R; ++i) column[c][i] = row[i]; }
      }
#pragma unroll 1
      for (uint32_t g = 0; g < n_aggregates; ++g) {
        const AggColumn& c = a.aggregates[g];
        const FusedInput& input = s_inputs[g];
        const uint32_t function = c.function;
        const bool has_value = function == HY_AGG_MIN || function == HY_AGG_MAX || function == HY_AGG_SUM || function == HY_AGG_AVG;
        uint64_t value[R];
        uint32_t nulls = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) value[i] = 0;
        if (uniform(input.n_nodes) && !(debug & 2)) evaluate_input(input, column, column_nulls, pass, value, &nulls);
        if (debug & 1) continue;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          if (!(((pass & ~nulls) >> i) & 1)) continue;   // NULL inputs leave the aggregate unchanged
          const uint64_t contribution = has_value ? contribution_from(c, value[i]) : 0;
          if ((in_global >> i) & 1) { merge_global(a, slot[i], g, contribution, 1); continue; }
          const bool is_dense = dense[i] < FUSED_DENSE;
          uint64_t* target = is_dense ? &s_cell_values[(dense[i] * (n_aggregates + 2) + g) * FUSED_CELLS + (lane % FUSED_CELLS)] : &s_values[slot[i] * n_aggregates + g];
          uint32_t* count = is_dense ? &s_cell_counts[(dense[i] * n_aggregates + g) * FUSED_CELLS + (lane % FUSED_CELLS)] : &s_counts[slot[i] * n_aggregates + g];
          if (has_value) accumulate_lds(c, target, contribution);
          atomicAdd(count, 1u);
        }
      }
    }
  }
  if (lane == 0 && wave_passed) atomicAdd(&s_n_groups[1], wave_passed);
  __syncthreads();
  if (tid == 0 && s_n_groups[1]) atomicAdd(reinterpret_cast<unsigned long long*>(a.overflow + FLAG_PASSED), static_cast<unsigned long long>(s_n_groups[1]));
  // the dense groups' cells -> their slots (one thread per group and accumulator; nobody else touches the table now)
  {
    const uint32_t n_dense = min(s_n_groups[0], FUSED_DENSE);
    for (uint32_t item = tid; item < n_dense * (n_aggregates + 2); item += 256) {
      const uint32_t d = item / (n_aggregates + 2), g = item % (n_aggregates + 2), slot = s_slot_of_dense[d];
      const uint64_t* cells = &s_cell_values[(d * (n_aggregates + 2) + g) * FUSED_CELLS];
      if (g < n_aggregates) {
        const AggColumn& c = a.aggregates[g];
        uint64_t value = s_values[slot * n_aggregates + g];
        uint32_t count = s_counts[slot * n_aggregates + g];
        for (uint32_t k = 0; k < FUSED_CELLS; ++k) {
          value = combine(c, value, cells[k]);
          count += s_cell_counts[(d * n_aggregates + g) * FUSED_CELLS + k];
        }
        s_values[slot * n_aggregates + g] = value;
        s_counts[slot * n_aggregates + g] = count;
      } else {
        uint64_t row = g == n_aggregates ? s_first[slot] : s_last[slot];
        for (uint32_t k = 0; k < FUSED_CELLS; ++k) row = g == n_aggregates ? min(row, cells[k]) : max(row, cells[k]);
        if (g == n_aggregates) s_first[slot] = row; else s_last[slot] = row;
      }
    }
  }
  __syncthreads();
  // the chunk's groups -> the global table
  for (uint32_t s = tid; s < slots; s += 256) {
    if (s_tags[s] == TAG_EMPTY) continue;
    if (__hip_atomic_load(&a.overflow[FLAG_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    uint64_t tuple[MAX_GROUPBY + 1];
    for (uint32_t w = 0; w < words; ++w) tuple[w] = s_keys[s * words + w];
    const uint32_t gslot = global_slot(a, tuple, words, hash_tuple(tuple, words));
    if (gslot == 0xFFFFFFFFu) { a.overflow[FLAG_OVERFLOW] = 1; continue; }
    atomicMin(reinterpret_cast<unsigned long long*>(&a.first_row[gslot]), static_cast<unsigned long long>(s_first[s]));
    atomicMax(reinterpret_cast<unsigned long long*>(&a.last_row[gslot]), static_cast<unsigned long long>(s_last[s]));
    for (uint32_t g = 0; g < n_aggregates; ++g) merge_global(a, gslot, g, s_values[s * n_aggregates + g], s_counts[s * n_aggregates + g]);
  }
}

#include "fused_small.hpp"

// (One atomic on the group counter per workgroup: one per group serialises at its L2 channel -- 100 000 groups took 0.36 ms that way.)
constexpr uint32_t COMPACT_THREADS = 1024;
__global__ __launch_bounds__(COMPACT_THREADS) void compact_groups(AggArgs a, uint32_t* counter, uint64_t* out_keys, uint64_t* out_first, uint64_t* out_last, uint64_t* out_values,
                                                                  uint64_t* out_counts, uint32_t out_capacity, StagedGroups staged) {
  __shared__ uint32_t s_wave_base[COMPACT_THREADS / 64 + 1];
  const uint32_t slot = blockIdx.x * COMPACT_THREADS + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool taken = slot < a.capacity && a.tags[slot] != TAG_EMPTY;
  const uint64_t peers = __ballot(taken);
  if (lane == 0) s_wave_base[wave] = static_cast<uint32_t>(__popcll(peers));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < COMPACT_THREADS / 64; ++w) { const uint32_t count = s_wave_base[w]; s_wave_base[w] = total; total += count; }
    s_wave_base[COMPACT_THREADS / 64] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  if (!taken) return;
  const uint32_t idx = s_wave_base[COMPACT_THREADS / 64] + s_wave_base[wave] + static_cast<uint32_t>(__popcll(peers & ((1ull << lane) - 1)));
  if (idx >= out_capacity) return;
  const uint32_t words = a.n_groupby + 1;
  const bool stage = idx < staged.capacity;
  for (uint32_t w = 0; w < words; ++w) {
    const uint64_t v = a.keys[static_cast<size_t>(slot) * words + w];
    out_keys[static_cast<size_t>(idx) * words + w] = v;
    if (stage) staged.keys[static_cast<size_t>(idx) * words + w] = v;
  }
  out_first[idx] = a.first_row[slot];
  out_last[idx] = a.last_row[slot];
  if (stage) { staged.first[idx] = a.first_row[slot]; staged.last[idx] = a.last_row[slot]; }
  for (uint32_t g = 0; g < a.n_aggregates; ++g) {
    const uint64_t v = a.values[static_cast<size_t>(slot) * a.n_aggregates + g], c = a.counts[static_cast<size_t>(slot) * a.n_aggregates + g];
    out_values[static_cast<size_t>(idx) * a.n_aggregates + g] = v;
    out_counts[static_cast<size_t>(idx) * a.n_aggregates + g] = c;
    if (stage) { staged.values[static_cast<size_t>(idx) * a.n_aggregates + g] = v; staged.counts[static_cast<size_t>(idx) * a.n_aggregates + g] = c; }
  }
}

// the flag words (FLAG_*) -> the header of the pinned block
__global__ void publish_group_flags(const uint32_t* flags, uint32_t* header) {
  header[0] = flags[0];
  header[1] = flags[1];
  header[2] = flags[2];
  header[3] = flags[3];
  header[4] = flags[4];   // (FLAG_PASSED, two words: fused_rows only)
  header[5] = flags[5];
  header[6] = flags[6];
  __threadfence_system();
}

// ANY(): value of the column at each group's representative row.
__global__ void gather_values(const DevSegment* segments, const hy_row_id* rows, uint32_t n, uint32_t is_float, uint64_t* bits, uint8_t* is_null) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Value v = column_value(segments, rows[i].chunk_id, rows[i].chunk_offset);
  is_null[i] = v.is_null;
  bits[i] = is_float ? static_cast<uint64_t>(__double_as_longlong(v.f)) : static_cast<uint64_t>(v.i);
}

// ---- large results are finished where they are ------------------------------------------------------------------------
// A result of 10^5 .. 10^7 groups used to come to the host as five arrays, be ordered there and be rewritten column by column: 3.5 of the
// 5 ms of a 100 000-group aggregate over 60 M rows, 220 of the 240 ms of a 4 M-group one.  The same steps as kernels: the extent of the
// keys (is the immediate-key shortcut taken?), one 32-bit sort key per group, the radix sort of join.hip, RowIDs and result columns written
// in result order -- and one copy per output array.
struct GroupExtent {
  unsigned long long key_min, key_max;   // over the groups with a value: biased key + 1 (as run_aggregate's host loop)
  unsigned long long row_max;            // largest first row
};
// (`partial`: device memory, {~0, 0, 0} and a zero counter behind it; the last workgroup publishes it into the pinned `out` -- atomics on
//  pinned host memory cross PCIe one by one.)
__global__ __launch_bounds__(256) void finish_extent(const uint64_t* keys, const uint64_t* first, uint32_t n_groups, uint32_t words, uint32_t int_key, GroupExtent* partial,
                                                     GroupExtent* out) {
  __shared__ unsigned long long s_min, s_max, s_row;
  if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0; s_row = 0; }
  __syncthreads();
  unsigned long long low = ~0ull, high = 0, row = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_groups; i += gridDim.x * 256) {
    row = first[i] > row ? first[i] : row;
    if (int_key && !(keys[size_t{i} * words] & 1)) {
      const unsigned long long k = static_cast<unsigned long long>(static_cast<int64_t>(keys[size_t{i} * words + 1]) - static_cast<int64_t>(INT32_MIN)) + 1;
      low = k < low ? k : low;
      high = k > high ? k : high;
    }
  }
  for (int step = 32; step > 0; step >>= 1) {   // (64-bit LDS atomics of every lane on three words: 1.3 ms for 100 000 groups)
    const unsigned long long other_low = __shfl_xor(low, step), other_high = __shfl_xor(high, step), other_row = __shfl_xor(row, step);
    low = other_low < low ? other_low : low;
    high = other_high > high ? other_high : high;
    row = other_row > row ? other_row : row;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&s_min, low);
    atomicMax(&s_max, high);
    atomicMax(&s_row, row);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&partial->key_min, s_min);
    atomicMax(&partial->key_max, s_max);
    atomicMax(&partial->row_max, s_row);
    __threadfence();
    if (atomicAdd(reinterpret_cast<uint32_t*>(partial + 1), 1u) + 1 == gridDim.x) {
      __threadfence();
      out->key_min = __hip_atomic_load(&partial->key_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out->key_max = __hip_atomic_load(&partial->key_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out->row_max = __hip_atomic_load(&partial->row_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
    }
  }
}

// sort key of group i: its first row, or (immediate keys) 0 for the NULL group and key - smallest key + 1 otherwise
__global__ __launch_bounds__(256) void finish_sort_keys(const uint64_t* keys, const uint64_t* first, uint32_t n_groups, uint32_t words, uint32_t immediate, uint64_t key_min,
                                                        uint32_t* sort_keys, uint32_t* ids) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_groups) return;
  uint32_t key = static_cast<uint32_t>(first[i]);
  if (immediate) {
    const uint64_t k = static_cast<uint64_t>(static_cast<int64_t>(keys[size_t{i} * words + 1]) - static_cast<int64_t>(INT32_MIN)) + 1;
    key = (keys[size_t{i} * words] & 1) ? 0u : static_cast<uint32_t>(k - key_min) + 1u;
  }
  sort_keys[i] = key;
  ids[i] = i;
}

// RowID of every result row's representative row (rows_of_groups: the groups' first rows, or their last ones: aggregate_hash.cpp:388-401)
__global__ __launch_bounds__(256) void finish_row_ids(const uint32_t* order, const uint64_t* rows_of_groups, uint32_t n_groups, const uint64_t* row_base, uint32_t n_chunks,
                                                      uint32_t uniform_size, hy_row_id* out) {
  const uint32_t o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n_groups) return;
  const uint64_t row = rows_of_groups[order[o]];
  uint32_t chunk;
  if (uniform_size) {
    chunk = static_cast<uint32_t>(row) / uniform_size;
    if (chunk >= n_chunks) chunk = n_chunks - 1;
  } else {
    uint32_t low = 0, high = n_chunks;   // the last chunk whose base is <= row
    while (high - low > 1) {
      const uint32_t middle = (low + high) / 2;
      if (row_base[middle] <= row) low = middle; else high = middle;
    }
    chunk = low;
  }
  out[o] = hy_row_id{chunk, static_cast<uint32_t>(row - row_base[chunk])};
}

// One result column in result order: COUNT / SUM / AVG / MIN / MAX from accumulator `primary` (the switch of run_aggregate's host loop).
struct FinishColumn {
  uint32_t function, out_type, is_float, primary;
};
__global__ __launch_bounds__(256) void finish_column(const uint32_t* order, const uint64_t* values, const uint64_t* counts, uint32_t n_groups, uint32_t n_device, FinishColumn c,
                                                     void* out_values, uint8_t* out_null) {
  const uint32_t o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n_groups) return;
  const size_t at = size_t{order[o]} * n_device + c.primary;
  const uint64_t bits = values[at], count = counts[at];
  bool is_null = count == 0;
  int64_t vi = 0;
  double vf = 0.0;
  switch (c.function) {
    case HY_AGG_COUNT: vi = static_cast<int64_t>(count); is_null = false; break;
    case HY_AGG_SUM:
      if (c.is_float) vf = __longlong_as_double(static_cast<long long>(bits)); else vi = static_cast<int64_t>(bits);
      break;
    case HY_AGG_AVG:
      if (count) vf = __longlong_as_double(static_cast<long long>(bits)) / static_cast<double>(count);   // aggregate_hash.cpp:166
      break;
    default:   // MIN / MAX
      if (c.is_float) {
        int64_t ordered = static_cast<int64_t>(bits);
        if (ordered < 0) ordered ^= 0x7FFFFFFFFFFFFFFFll;
        vf = __longlong_as_double(ordered);
      } else vi = static_cast<int64_t>(bits);
      break;
  }
  if (out_null) out_null[o] = is_null;
  switch (c.out_type) {
    case HY_TYPE_INT: static_cast<int32_t*>(out_values)[o] = is_null ? 0 : static_cast<int32_t>(vi); break;
    case HY_TYPE_LONG: static_cast<int64_t*>(out_values)[o] = is_null ? 0 : vi; break;
    case HY_TYPE_FLOAT: static_cast<float*>(out_values)[o] = is_null ? 0.f : static_cast<float>(vf); break;
    default: static_cast<double*>(out_values)[o] = is_null ? 0.0 : vf; break;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
static uint32_t result_type(uint32_t function, uint32_t input_type) {   // window_function_traits.hpp:11-77
  const bool is_float = input_type == HY_TYPE_FLOAT || input_type == HY_TYPE_DOUBLE;
  switch (function) {
    case HY_AGG_COUNT:
    case HY_AGG_COUNT_DISTINCT: return HY_TYPE_LONG;
    case HY_AGG_AVG:
    case HY_AGG_STDDEV_SAMP: return HY_TYPE_DOUBLE;
    case HY_AGG_SUM: return is_float ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
    default: return input_type;
  }
}

// Global row number -> RowID.  Chunks of one size (all but the last) are the rule: then the chunk is a multiplication by the
// reciprocal (+ a fix-up: the product is within one of the quotient) instead of a 64-bit division or a search per group.
struct RowIdOf {
  const hy_column* shape;
  uint64_t size = 0;       // common chunk size, 0: search
  double inverse = 0.0;
  explicit RowIdOf(const hy_column* column) : shape(column) {
    if (column->n_chunks > 1 && column->row_base[1] > 0 && column->rows < (1ull << 52)) {
      size = column->row_base[1];
      for (uint32_t c = 1; c < column->n_chunks && size; ++c) if (column->row_base[c] != c * size) size = 0;
      if (size) inverse = 1.0 / static_cast<double>(size);
    }
  }
  hy_row_id operator()(uint64_t global_row) const {
    if (size) {
      uint64_t chunk = static_cast<uint64_t>(static_cast<double>(global_row) * inverse);
      if (chunk * size > global_row) --chunk;
      else if ((chunk + 1) * size <= global_row) ++chunk;
      if (chunk >= shape->n_chunks) chunk = shape->n_chunks - 1;
      return hy_row_id{static_cast<uint32_t>(chunk), static_cast<uint32_t>(global_row - chunk * size)};
    }
    const auto it = std::upper_bound(shape->row_base.begin(), shape->row_base.end(), global_row);
    const uint32_t chunk = static_cast<uint32_t>(it - shape->row_base.begin()) - 1;
    return hy_row_id{chunk, static_cast<uint32_t>(global_row - shape->row_base[chunk])};
  }
};

// What the device table holds after one pass over the table: the groups' tuples, first / last rows and accumulators.
struct DeviceGroups {
  uint32_t n_groups = 0;
  uint64_t passed_rows = 0;   // fused_rows: rows that passed the filters
  std::vector<uint64_t> keys, first, last, values, counts;
  // Large results whose caller finishes them on the device (finish_groups_on_device): the same five arrays stay in device memory.
  bool keep_on_device = false;   // in: leave more than STAGED_GROUPS groups where they are
  bool on_device = false;        // out
  DeviceBuffer d_keys, d_first, d_last, d_values, d_counts;
  // in: the column whose aggregate_hint remembers the path (nullptr: none) and the signature of this GROUP BY
  const hy_column* hint_owner = nullptr;
  uint64_t hint_signature = 0;
};

// DeviceGroups::on_device -> the host vectors (the host finish after all)
static hy_status download_groups(DeviceGroups& groups, uint32_t words, uint32_t n_aggregates, hipStream_t stream) {
  const uint32_t n_groups = groups.n_groups;
  const size_t per_group = n_aggregates ? n_aggregates : 1;
  groups.keys.resize(size_t{n_groups} * words);
  groups.first.resize(n_groups);
  groups.last.resize(n_groups);
  groups.values.resize(n_groups * per_group);
  groups.counts.resize(n_groups * per_group);
  HY_HIP(hipMemcpyAsync(groups.keys.data(), groups.d_keys.ptr, 8 * groups.keys.size(), hipMemcpyDeviceToHost, stream));
  HY_HIP(hipMemcpyAsync(groups.first.data(), groups.d_first.ptr, 8 * size_t{n_groups}, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipMemcpyAsync(groups.last.data(), groups.d_last.ptr, 8 * size_t{n_groups}, hipMemcpyDeviceToHost, stream));
  if (n_aggregates) {
    HY_HIP(hipMemcpyAsync(groups.values.data(), groups.d_values.ptr, 8 * groups.values.size(), hipMemcpyDeviceToHost, stream));
    HY_HIP(hipMemcpyAsync(groups.counts.data(), groups.d_counts.ptr, 8 * groups.counts.size(), hipMemcpyDeviceToHost, stream));
  }
  HY_HIP(hipStreamSynchronize(stream));
  groups.on_device = false;
  return HY_OK;
}

// Runs aggregate_rows + compact_groups for the columns wired into `a` (GROUP BY and device accumulators), retrying with a
// larger global table when it overflows.
static uint64_t* g_agg_trace = nullptr;
static uint32_t g_agg_trace_slices = 0;

template <bool SCATTER, int WORDS, bool NARROW>
static void launch_partition_rows_as(uint32_t grid, size_t lds, hipStream_t stream, const AggArgs& a, const PartitionArgs& pa) {
  static OncePerDevice lds_raised;   // 2^14 partitions: 64 KiB of counters, the most a workgroup gets without asking
  uint64_t device_bit = 0;
  if (lds_raised.pending(&device_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(partition_rows<SCATTER, WORDS, NARROW>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    lds_raised.done(device_bit);
  }
  hipLaunchKernelGGL((partition_rows<SCATTER, WORDS, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa);
}
template <bool SCATTER, bool NARROW>
static void launch_partition_rows_of(uint32_t words, uint32_t grid, size_t lds, hipStream_t stream, const AggArgs& a, const PartitionArgs& pa) {
  switch (words) {
    case 1: launch_partition_rows_as<SCATTER, 1, NARROW>(grid, lds, stream, a, pa); break;
    case 2: launch_partition_rows_as<SCATTER, 2, NARROW>(grid, lds, stream, a, pa); break;
    case 3: launch_partition_rows_as<SCATTER, 3, NARROW>(grid, lds, stream, a, pa); break;
    case 4: launch_partition_rows_as<SCATTER, 4, NARROW>(grid, lds, stream, a, pa); break;
#if HY_MAX_GROUPBY > 4
    case 5: launch_partition_rows_as<SCATTER, 5, NARROW>(grid, lds, stream, a, pa); break;
    case 6: launch_partition_rows_as<SCATTER, 6, NARROW>(grid, lds, stream, a, pa); break;
    case 7: launch_partition_rows_as<SCATTER, 7, NARROW>(grid, lds, stream, a, pa); break;
    case 8: launch_partition_rows_as<SCATTER, 8, NARROW>(grid, lds, stream, a, pa); break;
#endif
    default: launch_partition_rows_as<SCATTER, MAX_GROUPBY + 1, NARROW>(grid, lds, stream, a, pa); break;
  }
}
static void launch_partition_rows(bool scatter, uint32_t words, uint32_t grid, size_t lds, hipStream_t stream, const AggArgs& a, const PartitionArgs& pa) {
  if (!scatter) launch_partition_rows_of<false, false>(words, grid, lds, stream, a, pa);   // (counting does not build records)
  else if (pa.narrow) launch_partition_rows_of<true, true>(words, grid, lds, stream, a, pa);
  else launch_partition_rows_of<true, false>(words, grid, lds, stream, a, pa);
}
template <bool NARROW>
static void launch_aggregate_partitions(uint32_t words, uint32_t grid, size_t lds, hipStream_t stream, const AggArgs& a, const PartitionArgs& pa, const DirectGroups& direct) {
  switch (words) {
    case 1: hipLaunchKernelGGL((aggregate_partitions<1, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
    case 2: hipLaunchKernelGGL((aggregate_partitions<2, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
    case 3: hipLaunchKernelGGL((aggregate_partitions<3, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
    case 4: hipLaunchKernelGGL((aggregate_partitions<4, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
#if HY_MAX_GROUPBY > 4
    case 5: hipLaunchKernelGGL((aggregate_partitions<5, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
    case 6: hipLaunchKernelGGL((aggregate_partitions<6, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
    case 7: hipLaunchKernelGGL((aggregate_partitions<7, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
    case 8: hipLaunchKernelGGL((aggregate_partitions<8, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
#endif
    default: hipLaunchKernelGGL((aggregate_partitions<MAX_GROUPBY + 1, NARROW>), dim3(grid), dim3(256), lds, stream, a, pa, direct); break;
  }
}

static uint32_t g_agg_path = 0;   // debug: 0 = aggregate_rows, otherwise the partition bits of the partitioned path (last call of this process)
static uint32_t g_agg_finished_on_device = 0;   // debug: 1 = the last call's result was ordered and written by the finish kernels
static uint32_t g_agg_small = 0;  // debug: 1 = the last call's groups came from aggregate_small_domain

static uint32_t fused_lds_slots(uint32_t n_groupby) { return n_groupby ? FUSED_LDS_SLOTS : 8u; }

// (`fused`: the device copy of a FusedPlan -- fused_rows takes the place of aggregate_rows; its accumulators' inputs are expressions, which
//  the partitioned path cannot carry.)
static hy_status device_groups(AggArgs a, const hy_column* shape, DeviceGroups& out, const FusedPlan* fused = nullptr, const SmallDomainPlan* small = nullptr,
                               const FusedSmallPlan* fused_small = nullptr) {
  hipStream_t stream = current_stream();
  const uint32_t words = a.n_groupby + 1;
  const uint32_t n_aggregates = a.n_aggregates;
  const size_t lds_bytes = size_t{LDS_SLOTS} * (8 * words + 16 + 12 * n_aggregates + 4 + 4) + 4 * DENSE_GROUPS + 64 + 64 + SLICE_ROWS + (a.pos_cache ? SLICE_ROWS : 0);   // (+ aggregate_rows' PosList offsets, one word per four rows)
  uint64_t two_per_row = 1024;
  while (two_per_row < 2 * shape->rows + 1024) two_per_row <<= 1;
  // the number of groups is not known: 64 Ki slots, then 2 Mi, then 32 Mi, then two slots per row (never overflows)
  const uint64_t ladder[4] = {std::max<uint64_t>(1u << 16, 2 * uint64_t{LDS_SLOTS}), 1u << 21, 1u << 25, two_per_row};
  int rung = 0;
  // The partitioned path (tables with many groups): entered when aggregate_rows gives up; 2^bits partitions of about 64 Ki rows,
  // then -- if even those hold more groups than a workgroup's table -- the most the partitioning kernels take.
  constexpr uint32_t MAX_PARTITION_BITS = 14;
  // (the seventeen-word build -- nine to sixteen GROUP BY columns -- keeps to aggregate_rows and the global table: the partitioning kernels
  //  are instantiated per tuple size, and eight more sizes of them for plans this rare would double the library's build time)
  const bool can_partition = !fused && a.n_groupby > 0 && shape->rows < (1ull << 32) && FIXED_AGG_PARTITIONS && MAX_GROUPBY <= 8;
  uint32_t first_bits = 6;
  while (first_bits < MAX_PARTITION_BITS && (shape->rows >> first_bits) > 65536) ++first_bits;
  uint32_t partition_bits = 0;   // 0: aggregate_rows
  if (can_partition && option(HY_OPT_AGG_PARTITION_BITS) > 0) partition_bits = std::min<uint32_t>(MAX_PARTITION_BITS, static_cast<uint32_t>(option(HY_OPT_AGG_PARTITION_BITS)));   // (tests: force the path)
  // A GROUP BY over the same columns ended on the partitioned path before: start there (the attempt aggregate_rows abandons at one row in
  // eight outside its tables costs about as much as the partitioned path itself; columns do not change, neither does the outcome).
  if (can_partition && partition_bits == 0 && !small && t_aggregate_next_path > 1) partition_bits = std::min<uint32_t>(MAX_PARTITION_BITS, t_aggregate_next_path - 1);   // (the caller knows how this GROUP BY went before)
  t_aggregate_next_path = 0;
  t_aggregate_recommended = 0;
  const bool hinted = can_partition && partition_bits == 0 && out.hint_owner && !small;
  if (hinted) {
    const uint64_t hint = out.hint_owner->aggregate_hint.load(std::memory_order_relaxed);
    if (hint >> 8 == out.hint_signature >> 8 && (hint & 0xFF) > 1) partition_bits = std::min<uint32_t>(MAX_PARTITION_BITS, static_cast<uint32_t>(hint & 0xFF) - 1);
  }
  bool unlimited = false, partitions_ready = false;
  DeviceBuffer part_offsets, part_rows, part_sums;
  PartitionArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  const bool timing = HY_DEBUG_ENV("HY_AGG_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what, int round) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[aggregate]   round %d %-18s +%8.3f ms\n", round, what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  for (int round = 0; round < 12; ++round) {
    if (partition_bits && rung == 0) rung = 1;   // many groups are a given on this path
    const uint64_t capacity = std::min(ladder[rung], two_per_row);
    if (capacity > (1ull << 31)) return fail(HY_ERR_UNSUPPORTED, "too many rows for the device group table");
    DeviceBuffer tags, keys, first, last, values, counts, flags, small_nibbles, small_slots;
    HY_TRY(tags.alloc(4 * capacity));
    HY_TRY(keys.alloc(8 * capacity * words));
    HY_TRY(first.alloc(8 * capacity));
    HY_TRY(last.alloc(8 * capacity));
    HY_TRY(values.alloc(8 * capacity * (n_aggregates ? n_aggregates : 1)));
    HY_TRY(counts.alloc(8 * capacity * (n_aggregates ? n_aggregates : 1)));
    HY_TRY(flags.alloc(64));
    HY_HIP(hipMemsetAsync(tags.ptr, 0, 4 * capacity, stream));
    HY_HIP(hipMemsetAsync(flags.ptr, 0, 64, stream));
    lap("table allocated", round);
    a.capacity = static_cast<uint32_t>(capacity);
    a.tags = tags.as<uint32_t>();
    a.keys = keys.as<uint64_t>();
    a.first_row = first.as<uint64_t>();
    a.last_row = last.as<uint64_t>();
    a.values = values.as<uint64_t>();
    a.counts = counts.as<uint64_t>();
    a.overflow = flags.as<uint32_t>();
    // rows outside the LDS tables cost a handful of device-scope atomics each; the partitioned path costs about as much as one
    // such row in sixteen -- plus the attempt it abandons: aggregate_rows goes on up to one row in eight (tables with thousands of
    // groups are recognised by their first slices, FLAG_CROWDED); partitions too coarse for their tables are refined at one in sixteen.
    // (No limit where there is nothing to switch to.)
    const bool last_resort = !can_partition || unlimited || (partition_bits && partition_bits >= MAX_PARTITION_BITS);
    // (SSB Q2.1 at SF30: 280 groups, 9 % of 1.4 M rows outside the 256-slot tables -- 3.9 ms with them, 4.5 ms through the give-up at rows / 16)
    const int spill_shift = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(8, option(HY_OPT_AGG_SPILL_SHIFT))));
    a.spill_limit = last_resort ? 0xFFFFFFFFu : static_cast<uint32_t>(std::max<uint64_t>(65536, shape->rows >> (partition_bits ? 4 : spill_shift)));
    a.trace = nullptr;
    if (HY_DEBUG_ENV("HY_AGG_TRACE") && shape->n_slices <= (1u << 14)) {
      static uint64_t* trace_buffer = nullptr;
      if (!trace_buffer) (void)hipMalloc(reinterpret_cast<void**>(&trace_buffer), 8 * 12 * size_t{1u << 14});
      (void)hipMemsetAsync(trace_buffer, 0, 8 * 12 * size_t{shape->n_slices}, stream);
      a.trace = trace_buffer;
      g_agg_trace = trace_buffer;
      g_agg_trace_slices = shape->n_slices;
    }
    // where the groups go, densely (compact_groups; aggregate_partitions with DirectGroups)
    DeviceBuffer c_keys, c_first, c_last, c_values, c_counts;
    // (the partitioned path's groups mostly bypass the global table: room for 8 Mi of them whatever its size)
    const uint32_t out_capacity = static_cast<uint32_t>(std::min<uint64_t>(partition_bits ? std::max<uint64_t>(capacity, 1u << 23) : capacity, shape->rows + 1));
    HY_TRY(c_keys.alloc(8 * size_t{out_capacity} * words));
    HY_TRY(c_first.alloc(8 * size_t{out_capacity}));
    HY_TRY(c_last.alloc(8 * size_t{out_capacity}));
    HY_TRY(c_values.alloc(8 * size_t{out_capacity} * (n_aggregates ? n_aggregates : 1)));
    HY_TRY(c_counts.alloc(8 * size_t{out_capacity} * (n_aggregates ? n_aggregates : 1)));
    // pinned block: header | keys | first | last | values | counts for up to STAGED_GROUPS groups
    constexpr uint32_t STAGED_GROUPS = 4096;
    const uint32_t per_group = n_aggregates ? n_aggregates : 1;
    const size_t staged_bytes = 64 + 8 * size_t{STAGED_GROUPS} * (words + 2 + 2 * per_group);
    void* pinned_host = nullptr;
    void* pinned_dev = nullptr;
    HY_TRY(pinned_staging(staged_bytes, &pinned_host, &pinned_dev));
    auto staged_arrays = [&](void* base) {
      StagedGroups g;
      g.keys = reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(base) + 64);
      g.first = g.keys + size_t{STAGED_GROUPS} * words;
      g.last = g.first + STAGED_GROUPS;
      g.values = g.last + STAGED_GROUPS;
      g.counts = g.values + size_t{STAGED_GROUPS} * per_group;
      g.capacity = STAGED_GROUPS;
      return g;
    };
    g_agg_path = partition_bits;
    g_agg_small = small && partition_bits == 0 && !fused ? 1u : 0u;
    g_agg_small = fused && fused_small ? 2u : g_agg_small;
    if (shape->n_slices && shape->rows && fused && fused_small) {   // the Q1 shape: one workgroup of 1024 threads per chunk (fused_small.hpp)
      profile_begin(stream, HY_KERNEL_AGGREGATE);
      static OncePerDevice raised;
      uint64_t device_bit = 0;
      if (raised.pending(&device_bit)) {
        HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_small_domain), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(fs_lds_bytes())));
        raised.done(device_bit);
      }
      hipLaunchKernelGGL(fused_small_domain, dim3(shape->n_chunks), dim3(FS_THREADS), fs_lds_bytes(), stream, a, fused, *fused_small, shape->n_chunks);
      profile_end(stream);
    } else if (shape->n_slices && shape->rows && fused) {   // one workgroup per chunk
      const size_t fused_lds = size_t{fused_lds_slots(a.n_groupby)} * (8 * words + 16 + 12 * n_aggregates + 4 + 4) + 64 + 2 * size_t{SLICE_ROWS} + sizeof(ColumnView) * FUSED_VIEWS +
                               sizeof(ScanJob) * HY_MAX_FILTERS + sizeof(FusedInput) * n_aggregates + size_t{FUSED_DENSE} * FUSED_CELLS * (8 * (n_aggregates + 2) + 4 * n_aggregates);
      profile_begin(stream, HY_KERNEL_AGGREGATE);
      if (fused_lds > 65536) {
        static OncePerDevice fused_lds_raised;
        uint64_t device_bit = 0;
        if (fused_lds_raised.pending(&device_bit)) {
          HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          fused_lds_raised.done(device_bit);
        }
      }
      hipLaunchKernelGGL(fused_rows, dim3(shape->n_chunks), dim3(256), fused_lds, stream, a, fused, shape->n_chunks);
      profile_end(stream);
    } else if (shape->n_slices && shape->rows && partition_bits == 0 && small) {   // a handful of groups over dictionary columns (aggregate_small.hpp)
      // two launches: the GROUP BY ids and 1-byte columns per chunk, then the 2-byte column per (chunk, part of the value-id range); between
      // them a nibble per row (its dense group) and the chunks' global-table slots
      SmallDomainPlan plan = *small;
      const bool has_wide = plan.n_columns > plan.n_narrow;
      if (has_wide) {
        HY_TRY(small_nibbles.alloc(size_t{shape->n_chunks} * SD_NIBBLE_BYTES));
        HY_TRY(small_slots.alloc(4 * size_t{shape->n_chunks} * SD_DENSE));
        plan.nibbles = small_nibbles.as<uint8_t>();
        plan.chunk_slots = small_slots.as<uint32_t>();
      }
      profile_begin(stream, HY_KERNEL_AGGREGATE);
      hipLaunchKernelGGL(sd_groups, dim3(shape->n_chunks), dim3(SD_THREADS), 0, stream, a, plan, shape->n_chunks);
      if (has_wide && !(plan.debug & 2)) {
        static OncePerDevice raised;
        uint64_t device_bit = 0;
        if (raised.pending(&device_bit)) {
          HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sd_wide), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sd_wide_lds_bytes())));
          raised.done(device_bit);
        }
        const uint32_t blocks = (shape->n_chunks + 7) / 8 * 8 * SD_WIDE_PARTS;
        hipLaunchKernelGGL(sd_wide, dim3(blocks), dim3(SD_WIDE_THREADS), sd_wide_lds_bytes(), stream, a, plan, shape->n_chunks);
      }
      profile_end(stream);
    } else if (shape->n_slices && shape->rows && partition_bits == 0) {
      profile_begin(stream, HY_KERNEL_AGGREGATE);
      if (lds_bytes > 65536) {   // (seventeen-word tuples: 256 slots of them exceed what a workgroup gets without asking)
        static OncePerDevice rows_lds_raised;
        uint64_t device_bit = 0;
        if (rows_lds_raised.pending(&device_bit)) {
          HY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(aggregate_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          rows_lds_raised.done(device_bit);
        }
      }
      hipLaunchKernelGGL(aggregate_rows, dim3(shape->n_slices), dim3(256), lds_bytes, stream, a);
      profile_end(stream);
    } else if (shape->n_slices && shape->rows) {
      const uint32_t partitions = 1u << partition_bits;
      profile_begin(stream, HY_KERNEL_AGGREGATE);
      if (!partitions_ready) {
        pa.tile_slices = partition_bits <= 11 ? 1 : 8;   // (a tile's histogram is written and scanned: 2^bits cells per tile)
        pa.n_slices = shape->n_slices;
        pa.n_parts = (shape->n_slices + pa.tile_slices - 1) / pa.tile_slices;
        const uint64_t cells = uint64_t{partitions} * pa.n_parts;
        const uint32_t n_blocks = static_cast<uint32_t>((cells + SCAN_BLOCK - 1) / SCAN_BLOCK);
        HY_TRY(part_offsets.alloc(4 * cells));
        HY_TRY(part_sums.alloc(4 * (size_t{n_blocks} + 1)));
        uint32_t carried = 0;
        while (carried < n_aggregates && a.aggregates[carried].segments) ++carried;   // (run_aggregate puts COUNT(*) last)
        for (uint32_t g = carried; g < n_aggregates; ++g) if (a.aggregates[g].segments) return fail(HY_ERR_DEVICE, "aggregates with a column must come first (internal error)");
        pa.carried = carried;
        pa.record_words = (1 + words + carried + 1) / 2 * 2;
        pa.narrow = 1;   // every key and every carried input a 4-byte type: 32-bit words (partition_rows)
        for (uint32_t g = 0; g + 1 < words; ++g) if (a.groupby[g].data_type != HY_TYPE_INT && a.groupby[g].data_type != HY_TYPE_FLOAT) pa.narrow = 0;
        for (uint32_t g = 0; g < carried; ++g) if (a.aggregates[g].data_type != HY_TYPE_INT && a.aggregates[g].data_type != HY_TYPE_FLOAT) pa.narrow = 0;
        if (pa.narrow) pa.record_words = (2 + (words - 1) + carried + 3) / 4 * 2;   // (16-byte units either way)
        HY_TRY(part_rows.alloc(8 * size_t{pa.record_words} * shape->rows));
        pa.bits = partition_bits;
        pa.offsets = part_offsets.as<uint32_t>();
        pa.records = part_rows.as<uint64_t>();
        pa.total_rows = shape->rows;
        const size_t per_slot = 8 * words + 16 + 12 * n_aggregates + 4;
        pa.lds_slots = 2048;
        const size_t lds_budget = static_cast<size_t>(std::max<int64_t>(1024, FIXED_AGG_LDS_BUDGET));
        while (pa.lds_slots > 64 && pa.lds_slots * per_slot > lds_budget) pa.lds_slots >>= 1;
        launch_partition_rows(false, words, pa.n_parts, 4 * size_t{partitions}, stream, a, pa);
        hipLaunchKernelGGL(scan_blocks, dim3(n_blocks), dim3(256), 0, stream, pa.offsets, cells, part_sums.as<uint32_t>());
        hipLaunchKernelGGL(scan_sums, dim3(1), dim3(256), 0, stream, part_sums.as<uint32_t>(), n_blocks);
        hipLaunchKernelGGL(scan_add, dim3(n_blocks), dim3(256), 0, stream, pa.offsets, cells, part_sums.as<uint32_t>());
        launch_partition_rows(true, words, pa.n_parts, 4 * size_t{partitions}, stream, a, pa);
        partitions_ready = true;
      }
      const size_t per_slot = 8 * words + 16 + 12 * n_aggregates + 4;
      // about 16 Ki rows per workgroup: coarse partitions (long contiguous runs for the scatter) are shared by several
      pa.split = 1;
      while (pa.split < 64 && (shape->rows >> partition_bits) / pa.split > 16384) pa.split <<= 1;
      if (FIXED_AGG_SPLIT > 0) pa.split = static_cast<uint32_t>(FIXED_AGG_SPLIT);
      DirectGroups direct;
      std::memset(&direct, 0, sizeof(direct));
      direct.enabled = pa.split == 1 ? 1u : 0u;
      direct.out_capacity = out_capacity;
      direct.counter = flags.as<uint32_t>() + FLAG_GROUPS;
      direct.keys = c_keys.as<uint64_t>();
      direct.first = c_first.as<uint64_t>();
      direct.last = c_last.as<uint64_t>();
      direct.values = c_values.as<uint64_t>();
      direct.counts = c_counts.as<uint64_t>();
      direct.staged = staged_arrays(pinned_dev);
      if (pa.narrow) launch_aggregate_partitions<true>(words, partitions * pa.split, pa.lds_slots * per_slot + 64, stream, a, pa, direct);
      else launch_aggregate_partitions<false>(words, partitions * pa.split, pa.lds_slots * per_slot + 64, stream, a, pa, direct);
      profile_end(stream);
    }
    lap("kernels launched", round);
    uint32_t host_flags[7] = {0, 0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(compact_groups, dim3(static_cast<uint32_t>((capacity + COMPACT_THREADS - 1) / COMPACT_THREADS)), dim3(COMPACT_THREADS), 0, stream, a, flags.as<uint32_t>() + FLAG_GROUPS, c_keys.as<uint64_t>(),
                       c_first.as<uint64_t>(), c_last.as<uint64_t>(), c_values.as<uint64_t>(), c_counts.as<uint64_t>(), out_capacity, staged_arrays(pinned_dev));
    hipLaunchKernelGGL(publish_group_flags, dim3(1), dim3(1), 0, stream, flags.as<uint32_t>(), static_cast<uint32_t*>(pinned_dev));
    lap("compact launched", round);
    HY_HIP(hipStreamSynchronize(stream));
    lap("device finished", round);
    std::memcpy(host_flags, pinned_host, 28);
    if (host_flags[FLAG_SMALL_REFUSED] && small) {   // (a value met sixteen times in one group of one chunk)
      small = nullptr;
      continue;
    }
    if (host_flags[FLAG_SMALL_REFUSED] && fused_small) {   // (a fifth group in a chunk, a NULL input: fused_rows)
      fused_small = nullptr;
      continue;
    }
    if (host_flags[FLAG_GIVE_UP]) {   // too many rows outside the LDS tables: partition (more finely)
      if (partition_bits == 0) partition_bits = first_bits;
      else if (partition_bits < MAX_PARTITION_BITS) partition_bits = MAX_PARTITION_BITS;
      else unlimited = true;
      partitions_ready = false;
      continue;
    }
    if (host_flags[FLAG_OVERFLOW]) {   // table overflow: retry with a larger one
      if (rung < 3) ++rung;
      else return fail(HY_ERR_DEVICE, "the device group table overflowed at two slots per row (internal error)");
      continue;
    }
    const uint32_t n_groups = host_flags[FLAG_GROUPS];
    if (n_groups > out_capacity) {   // (direct groups + the global table's: more than the arrays hold)
      if (rung < 3) ++rung;
      else return fail(HY_ERR_DEVICE, "more groups than rows (internal error)");
      continue;
    }
    out.n_groups = n_groups;
    out.passed_rows = static_cast<uint64_t>(host_flags[5]) << 32 | host_flags[4];
    // What the next GROUP BY over these columns should start with: the path this one ended on -- or, where aggregate_rows ran to the end over
    // a few million rows with more groups than a workgroup's table has slots (the rows of the groups that found no slot met in the global
    // table, atomic by atomic: SSB Q2.1's 280 groups over 1.4 M rows took 0.42 ms there and take 0.2 ms in sixteen partitions), 4 bits.
    uint32_t recommended_bits = partition_bits;
    if (partition_bits == 0 && can_partition && !small && n_groups > LDS_SLOTS && shape->rows < (8u << 20)) recommended_bits = 4;
    t_aggregate_recommended = recommended_bits ? recommended_bits + 1 : 0;
    if (out.hint_owner && can_partition && !small && option(HY_OPT_AGG_PARTITION_BITS) <= 0) out.hint_owner->aggregate_hint.store((out.hint_signature >> 8) << 8 | (recommended_bits + 1), std::memory_order_relaxed);
    if (out.keep_on_device && n_groups > STAGED_GROUPS) {
      std::swap(out.d_keys.ptr, c_keys.ptr);     std::swap(out.d_keys.capacity, c_keys.capacity);
      std::swap(out.d_first.ptr, c_first.ptr);   std::swap(out.d_first.capacity, c_first.capacity);
      std::swap(out.d_last.ptr, c_last.ptr);     std::swap(out.d_last.capacity, c_last.capacity);
      std::swap(out.d_values.ptr, c_values.ptr); std::swap(out.d_values.capacity, c_values.capacity);
      std::swap(out.d_counts.ptr, c_counts.ptr); std::swap(out.d_counts.capacity, c_counts.capacity);
      out.on_device = true;
      lap("groups stay", round);
      return HY_OK;
    }
    out.keys.resize(size_t{n_groups} * words);
    out.first.resize(n_groups);
    out.last.resize(n_groups);
    out.values.resize(size_t{n_groups} * (n_aggregates ? n_aggregates : 1));
    out.counts.resize(size_t{n_groups} * (n_aggregates ? n_aggregates : 1));
    if (n_groups && n_groups <= STAGED_GROUPS) {   // everything is already here
      const StagedGroups staged = staged_arrays(pinned_host);
      std::memcpy(out.keys.data(), staged.keys, 8 * out.keys.size());
      std::memcpy(out.first.data(), staged.first, 8 * size_t{n_groups});
      std::memcpy(out.last.data(), staged.last, 8 * size_t{n_groups});
      if (n_aggregates) {
        std::memcpy(out.values.data(), staged.values, 8 * out.values.size());
        std::memcpy(out.counts.data(), staged.counts, 8 * out.counts.size());
      }
    } else if (n_groups) {
      HY_HIP(hipMemcpyAsync(out.keys.data(), c_keys.ptr, 8 * out.keys.size(), hipMemcpyDeviceToHost, stream));
      HY_HIP(hipMemcpyAsync(out.first.data(), c_first.ptr, 8 * size_t{n_groups}, hipMemcpyDeviceToHost, stream));
      HY_HIP(hipMemcpyAsync(out.last.data(), c_last.ptr, 8 * size_t{n_groups}, hipMemcpyDeviceToHost, stream));
      if (n_aggregates) {
        HY_HIP(hipMemcpyAsync(out.values.data(), c_values.ptr, 8 * out.values.size(), hipMemcpyDeviceToHost, stream));
        HY_HIP(hipMemcpyAsync(out.counts.data(), c_counts.ptr, 8 * out.counts.size(), hipMemcpyDeviceToHost, stream));
      }
      HY_HIP(hipStreamSynchronize(stream));
    }
    lap("groups copied", round);
    return HY_OK;
  }
  return fail(HY_ERR_DEVICE, "the device group table did not settle (internal error)");
}


// Some value of a column (its first non-NULL one among the first rows of the chunks), as a double: STDDEV_SAMP accumulates
// sum(x - pivot) and sum((x - pivot)^2).  The reference runs Welford's recurrence (abstract_aggregate_operator.hpp:83-113);
// plain sums of x and x^2 cancel catastrophically when |mean| >> spread (values 1e9 + {0, 1, 2}: x^2 ~ 1e18, one ulp 128),
// shifted by a value of the data they do not: the variance formula is shift-invariant and the shifted terms are of the
// size of the spread.  out[0] = pivot, out[1] = 1 if one was found.
__global__ __launch_bounds__(256) void column_pivot(const DevSegment* segments, uint32_t n_chunks, double* out) {
  __shared__ double s_value;
  __shared__ uint32_t s_found;
  if (threadIdx.x == 0) s_found = 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t chunk = 0; chunk < n_chunks; ++chunk) {
    const uint32_t size = segments[chunk].size;
    for (uint32_t begin = 0; begin < size && begin < 16384; begin += 256) {
      const uint32_t row = begin + threadIdx.x;
      Value v{true, 0, 0.0};
      if (row < size) v = column_value(segments, chunk, row);
      if (!v.is_null) atomicMin(&s_found, threadIdx.x);
      __syncthreads();
      if (s_found != 0xFFFFFFFFu) {
        if (threadIdx.x == s_found) {
          const uint32_t type = segments[chunk].data_type;
          s_value = (type == HY_TYPE_FLOAT || type == HY_TYPE_DOUBLE) ? v.f : static_cast<double>(v.i);
        }
        __syncthreads();
        if (threadIdx.x == 0) { out[0] = s_value; out[1] = 1.0; }
        return;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) { out[0] = 0.0; out[1] = 0.0; }
}

// Host loops over the result's groups: disjoint output ranges, so large results (the partitioned path hands back 10^5 .. 10^7
// groups) are split over a few threads.
template <typename Body>
static void for_each_group_range(uint32_t n, Body body) {
  const uint32_t threads = n < (1u << 19) ? 1u : std::min<uint32_t>(8, std::max(1u, std::thread::hardware_concurrency()));
  if (threads <= 1) { body(0u, n); return; }
  std::vector<std::thread> workers;
  const uint32_t per = (n + threads - 1) / threads;
  for (uint32_t t = 0; t < threads; ++t) {
    const uint32_t begin = std::min(n, t * per), end = std::min(n, begin + per);
    if (begin < end) workers.emplace_back([=] { body(begin, end); });
  }
  for (auto& worker : workers) worker.join();
}

// hy_scan_project_aggregate: the filters and the aggregates' input expressions, checked and typed by run_fused below.  The aggregate
// specs then carry no columns -- input g is `inputs[g]` (n_nodes 0: COUNT(*)); inputs with the same `same_as` are one expression.
struct FusedQuery {
  const hy_column* shape;
  const hy_filter* filters;
  uint32_t n_filters;
  const hy_column* columns[FUSED_COLUMNS];   // the distinct columns the expressions read (FusedNode::column indexes them)
  uint32_t n_columns;
  FusedInput inputs[MAX_AGGREGATES];
  uint32_t same_as[MAX_AGGREGATES];
};

static hy_status run_aggregate(const hy_column* const* groupby, uint32_t n_groupby, const hy_aggregate_spec* specs, uint32_t n_aggregates,
                               hy_aggregate_result* result, const FusedQuery* fused = nullptr) {
  if (n_groupby > MAX_GROUPBY) return fail(HY_ERR_UNSUPPORTED, "more than %u GROUP BY columns stay on the CPU path", MAX_GROUPBY);
  if (n_aggregates > MAX_AGGREGATES) return fail(HY_ERR_UNSUPPORTED, "more than %u aggregates stay on the CPU path", MAX_AGGREGATES);
  const hy_column* shape = fused ? fused->shape : n_groupby ? groupby[0] : nullptr;
  for (uint32_t g = 0; g < n_aggregates && !shape; ++g) shape = specs[g].column;
  if (!shape) return fail(HY_ERR_INVALID, "hy_aggregate_hash needs at least one column (pass any column of the table for a lone COUNT(*))");
  // what an aggregate reads: a column, or (fused) an expression -- known to the accumulators only by its type and by a non-null marker
  static const DevSegment expression_marker{};
  auto has_input = [&](uint32_t g) { return fused ? fused->inputs[g].n_nodes != 0 : specs[g].column != nullptr; };
  auto input_type = [&](uint32_t g) -> uint32_t { return fused ? (fused->inputs[g].n_nodes ? fused->inputs[g].type : static_cast<uint32_t>(HY_TYPE_LONG))
                                                                : (specs[g].column ? specs[g].column->data_type : static_cast<uint32_t>(HY_TYPE_LONG)); };
  auto same_input = [&](uint32_t x, uint32_t y) { return fused ? fused->same_as[x] == fused->same_as[y] : specs[x].column == specs[y].column; };
  auto same_shape = [&](const hy_column* c) {
    if (c->n_chunks != shape->n_chunks || c->is_mvcc || (c->ref && c->ref->is_mvcc)) return false;
    for (uint32_t k = 0; k < c->n_chunks; ++k) if (c->host_segments[k].size != shape->host_segments[k].size) return false;
    return true;
  };
  bool reads_references = false;
  auto wire = [&reads_references](AggColumn& slot, const hy_column* column, uint32_t function) {
    if (column && column->is_reference) reads_references = true;
    slot.segments = column ? column->d_segments : nullptr;
    slot.data_type = column ? column->data_type : static_cast<uint32_t>(HY_TYPE_LONG);
    slot.is_float = column && (column->data_type == HY_TYPE_FLOAT || column->data_type == HY_TYPE_DOUBLE);
    slot.function = function;
  };
  auto wire_expression = [&](AggColumn& slot, uint32_t g, uint32_t function) {
    slot.segments = has_input(g) ? &expression_marker : nullptr;   // (never dereferenced: fused_rows evaluates FusedPlan::inputs)
    slot.data_type = input_type(g);
    slot.is_float = has_input(g) && (slot.data_type == HY_TYPE_FLOAT || slot.data_type == HY_TYPE_DOUBLE);
    slot.function = function;
  };
  AggArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_groupby = n_groupby;
  for (uint32_t g = 0; g < n_groupby; ++g) {
    if (!groupby[g] || !same_shape(groupby[g])) return fail(HY_ERR_INVALID, "GROUP BY column %u does not have the table's chunk layout", g);
    if (groupby[g]->data_type == HY_TYPE_STRING) return fail(HY_ERR_UNSUPPORTED, "string GROUP BY columns must be passed as dictionary segments of int64 key names (INTEGRATION.md)");
    wire(a.groupby[g], groupby[g], 0);
  }
  // Device accumulators.  STDDEV_SAMP takes two (sum of x - pivot as double + count, sum of (x - pivot)^2); COUNT(DISTINCT) takes none: it
  // is a second grouping by (GROUP BY columns, aggregate column) whose groups are counted per outer group.
  std::vector<int> primary(n_aggregates, -1), secondary(n_aggregates, -1);
  uint32_t n_device = 0;
  uint32_t spec_of_device[MAX_AGGREGATES] = {0, 0, 0, 0, 0, 0, 0, 0};   // (fused: whose expression feeds the accumulator)
  for (uint32_t g = 0; g < n_aggregates; ++g) {
    const hy_aggregate_spec& spec = specs[g];
    if (spec.function > HY_AGG_ANY) return fail(HY_ERR_INVALID, "unknown aggregate function %u", spec.function);
    if (!has_input(g) && spec.function != HY_AGG_COUNT) return fail(HY_ERR_INVALID, "only COUNT may omit its column (aggregate_hash.cpp:1002)");
    if (fused && spec.function > HY_AGG_COUNT) return fail(HY_ERR_UNSUPPORTED, "hy_scan_project_aggregate: MIN / MAX / SUM / AVG / COUNT only -- run the operator chain for function %u", spec.function);
    if (spec.column) {
      if (!same_shape(spec.column)) return fail(HY_ERR_INVALID, "aggregate column %u does not have the table's chunk layout", g);
      if (spec.column->data_type == HY_TYPE_STRING && spec.function != HY_AGG_COUNT) return fail(HY_ERR_UNSUPPORTED, "string aggregates stay on the CPU path");
    }
    if (spec.function == HY_AGG_COUNT_DISTINCT) {
      if (n_groupby + 1 > MAX_GROUPBY) return fail(HY_ERR_UNSUPPORTED, "COUNT(DISTINCT) with %u GROUP BY columns stays on the CPU path", n_groupby);
      continue;
    }
    // SUM and AVG of one floating-point column are the same accumulator (a double sum in row order and a count of the
    // non-NULL rows): TPC-H Q1 asks for both of l_quantity and of l_extendedprice.  (Integer columns: SUM adds int64,
    // AVG adds doubles -- two accumulators.)
    const bool float_column = has_input(g) && (input_type(g) == HY_TYPE_FLOAT || input_type(g) == HY_TYPE_DOUBLE);
    if (float_column && (spec.function == HY_AGG_SUM || spec.function == HY_AGG_AVG)) {
      for (uint32_t earlier = 0; earlier < g && primary[g] < 0; ++earlier) {
        if (has_input(earlier) && same_input(earlier, g) && (specs[earlier].function == HY_AGG_SUM || specs[earlier].function == HY_AGG_AVG)) primary[g] = primary[earlier];
      }
      if (primary[g] >= 0) continue;
    }
    const uint32_t wanted = spec.function == HY_AGG_STDDEV_SAMP ? 2 : 1;
    if (n_device + wanted > MAX_AGGREGATES) return fail(HY_ERR_UNSUPPORTED, "more than %u device accumulators stay on the CPU path", MAX_AGGREGATES);
    primary[g] = static_cast<int>(n_device);
    spec_of_device[n_device] = g;
    if (fused) {
      wire_expression(a.aggregates[n_device++], g, spec.function);
      continue;
    }
    wire(a.aggregates[n_device++], spec.column, spec.function == HY_AGG_STDDEV_SAMP ? AGG_SUM_SHIFTED : spec.function);
    if (spec.function == HY_AGG_STDDEV_SAMP) {
      secondary[g] = static_cast<int>(n_device);
      wire(a.aggregates[n_device++], spec.column, AGG_SUM_SQUARES);
      double* pivot_host = nullptr;   // a value of the column to shift by (column_pivot above)
      double* pivot_dev = nullptr;
      HY_TRY(pinned_staging(16, reinterpret_cast<void**>(&pivot_host), reinterpret_cast<void**>(&pivot_dev)));
      pivot_host[0] = pivot_host[1] = 0.0;
      if (spec.column->n_chunks) {
        hipLaunchKernelGGL(column_pivot, dim3(1), dim3(256), 0, current_stream(), spec.column->d_segments, spec.column->n_chunks, pivot_dev);
        HY_HIP(hipStreamSynchronize(current_stream()));
      }
      a.aggregates[n_device - 2].pivot = a.aggregates[n_device - 1].pivot = pivot_host[0];
    }
  }
  {   // accumulators with a column first, COUNT(*) last: the partitioned path carries the contributions of a prefix
    AggColumn wired[MAX_AGGREGATES];
    uint32_t place[MAX_AGGREGATES], next = 0;
    for (uint32_t d = 0; d < n_device; ++d) wired[d] = a.aggregates[d];
    for (uint32_t d = 0; d < n_device; ++d) if (wired[d].segments) place[d] = next++;
    for (uint32_t d = 0; d < n_device; ++d) if (!wired[d].segments) place[d] = next++;
    for (uint32_t d = 0; d < n_device; ++d) a.aggregates[place[d]] = wired[d];
    {
      uint32_t moved[MAX_AGGREGATES];
      for (uint32_t d = 0; d < n_device; ++d) moved[place[d]] = spec_of_device[d];
      for (uint32_t d = 0; d < n_device; ++d) spec_of_device[d] = moved[d];
    }
    for (uint32_t g = 0; g < n_aggregates; ++g) {
      if (primary[g] >= 0) primary[g] = static_cast<int>(place[primary[g]]);
      if (secondary[g] >= 0) secondary[g] = static_cast<int>(place[secondary[g]]);
    }
  }
  a.n_aggregates = n_device;
  a.slices = shape->d_slices;
  a.row_base = shape->d_row_base;
  hipStream_t stream = current_stream();
  const uint32_t words = n_groupby + 1;

  const bool timing = HY_DEBUG_ENV("HY_AGG_TIMING") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) std::fprintf(stderr, "[aggregate] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  auto dictionary_column = [&](const hy_column* column, uint32_t* width) {   // every chunk a dictionary segment with its values on the device, one id width, aligned
    if (!column || column->is_reference || column->has_dictionary_without_values || column->n_chunks == 0) return false;
    *width = column->host_segments[0].width;
    if (*width != 1 && *width != 2) return false;
    for (uint32_t k = 0; k < column->n_chunks; ++k) {
      const hy_segment& seg = column->host_segments[k];
      if (seg.encoding != HY_ENC_DICTIONARY || seg.width != *width || (!seg.aux && seg.aux_size) || reinterpret_cast<uintptr_t>(seg.data) % 16 != 0 || (*width == 1 && seg.aux_size > 255)) return false;
    }
    return true;
  };
  // every chunk a dictionary segment, one width of 1 or 2 bytes per value id, aligned (the dictionary's values may be anywhere: filters test ids)
  auto value_id_column = [&](const hy_column* column, uint32_t* width) {
    if (!column || column->is_reference || column->n_chunks == 0) return false;
    *width = column->host_segments[0].width;
    if (*width != 1 && *width != 2) return false;
    for (uint32_t k = 0; k < column->n_chunks; ++k) {
      const hy_segment& seg = column->host_segments[k];
      if (seg.encoding != HY_ENC_DICTIONARY || seg.width != *width || reinterpret_cast<uintptr_t>(seg.data) % 16 != 0) return false;
    }
    return true;
  };
  auto few_codes = [&]() {   // the GROUP BY columns of the small-domain kernels: 1-byte value ids, at most SD_CODES combinations per chunk
    if (n_groupby > SD_KEYS) return false;
    for (uint32_t g = 0; g < n_groupby; ++g) {
      uint32_t key_width = 0;
      if (!dictionary_column(groupby[g], &key_width) || key_width != 1) return false;
    }
    for (uint32_t k = 0; k < shape->n_chunks; ++k) {
      uint64_t product = 1;
      for (uint32_t g = 0; g < n_groupby; ++g) product *= uint64_t{groupby[g]->host_segments[k].aux_size} + 1;
      if (product > SD_CODES) return false;
    }
    return true;
  };
  DeviceGroups main_groups;
  {   // a large result of plain aggregates is finished on the device (finish kernels above); everything else comes to the host
    bool plain_functions = n_groupby > 0 && result->mem == HY_MEM_HOST && shape->rows < (1ull << 32);
    for (uint32_t g = 0; g < n_aggregates && plain_functions; ++g) {
      const uint32_t f = specs[g].function;
      plain_functions = primary[g] >= 0 && (f == HY_AGG_COUNT || f == HY_AGG_SUM || f == HY_AGG_AVG || f == HY_AGG_MIN || f == HY_AGG_MAX) && result->columns[g].values;
    }
    main_groups.keep_on_device = plain_functions;
    if (!fused && n_groupby) {
      uint64_t signature = 0x9E3779B97F4A7C15ull * (n_groupby + 1);
      for (uint32_t g = 0; g < n_groupby; ++g) signature = (signature ^ reinterpret_cast<uintptr_t>(groupby[g])) * 0xD6E8FEB86659FD93ull;
      main_groups.hint_owner = groupby[0];
      main_groups.hint_signature = signature | 0x100;   // (never zero above the path byte)
    }
  }
  if (fused) {
    // the plan in device memory: per filter the chunk jobs (prepare_jobs, like hy_table_scan), per accumulator its input expression
    const uint32_t n_chunks = shape->n_chunks;
    std::vector<size_t> staging_at(fused->n_filters + 1, 0);
    for (uint32_t f = 0; f < fused->n_filters; ++f) {
      const bool validate = fused->filters[f].predicate.condition == HY_FILTER_VALIDATE;
      staging_at[f + 1] = staging_at[f] + (validate ? 256 : align_up(scan_jobs_staging_bytes(fused->filters[f].column, &fused->filters[f].predicate), 256));
    }
    const size_t jobs_bytes = align_up(sizeof(ScanJob) * (size_t{n_chunks} + 1), 256);
    DeviceBuffer plan_buffer;
    HY_TRY(plan_buffer.alloc(align_up(sizeof(FusedPlan), 256) + jobs_bytes * fused->n_filters + staging_at[fused->n_filters] + 256));
    char* base = plan_buffer.as<char>();
    char* jobs_base = base + align_up(sizeof(FusedPlan), 256);
    char* staging_base = jobs_base + jobs_bytes * fused->n_filters;
    FusedPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    plan.n_filters = fused->n_filters;
    for (uint32_t f = 0; f < fused->n_filters; ++f) {
      ScanJob* jobs = reinterpret_cast<ScanJob*>(jobs_base + jobs_bytes * f);
      const hy_predicate& predicate = fused->filters[f].predicate;
      if (predicate.condition == HY_FILTER_VALIDATE) HY_TRY(prepare_visibility_scan_jobs(fused->filters[f].column, predicate.value.value_id, predicate.value2.value_id, predicate.column_is_nullable, jobs, staging_base + staging_at[f]));
      else HY_TRY(prepare_scan_jobs(fused->filters[f].column, &predicate, jobs, staging_base + staging_at[f]));
      plan.filters[f].segments = fused->filters[f].column->d_segments;
      plan.filters[f].jobs = jobs;
    }
    plan.n_columns = fused->n_columns;
    plan.lds_slots = fused_lds_slots(n_groupby);
    plan.debug = HY_DEBUG_ENV("HY_FUSED_DEBUG") ? static_cast<uint32_t>(atoi(HY_DEBUG_ENV("HY_FUSED_DEBUG"))) : 0u;
    for (uint32_t c = 0; c < fused->n_columns; ++c) plan.columns[c] = fused->columns[c]->d_segments;
    for (uint32_t d = 0; d < n_device; ++d) plan.inputs[d] = fused->inputs[spec_of_device[d]];
    HY_HIP(hipMemcpyAsync(base, &plan, sizeof(plan), hipMemcpyHostToDevice, stream));   // (`plan` lives until device_groups below has waited for the stream)
    // The TPC-H Q1 shape -- a handful of groups, filters on value ids, float expressions over dictionary columns -- has a kernel of its
    // own (fused_small.hpp); everything else, and whatever that kernel refuses at run time, takes fused_rows.
    FusedSmallPlan small_plan;
    std::memset(&small_plan, 0, sizeof(small_plan));
    for (uint32_t c = 0; c < FS_COLUMNS; ++c) small_plan.column_of_slot[c] = 0xFFFFFFFFu;
    bool lean = option(HY_OPT_FUSED_SMALL_DOMAIN) && shape->rows > 0 && fused->n_filters <= FS_FILTERS && fused->n_columns <= FS_COLUMNS && few_codes();
    for (uint32_t k = 0; k < shape->n_chunks && lean; ++k) lean = shape->host_segments[k].size <= FS_SPAN;
    for (uint32_t f = 0; f < fused->n_filters && lean; ++f) {
      const uint32_t condition = fused->filters[f].predicate.condition;
      lean = condition != HY_FILTER_VALIDATE && !(condition >= HY_PRED_IN && condition <= HY_PRED_NOT_LIKE_INSENSITIVE) && value_id_column(fused->filters[f].column, &small_plan.filter_width[f]);
    }
    uint32_t n_narrow = 0;
    for (uint32_t c = 0; c < fused->n_columns && lean; ++c) {   // 1-byte value ids: slots 0 .. FS_NARROW - 1; 2-byte value ids: the last slot
      uint32_t width = 0;
      lean = fused->columns[c]->data_type == HY_TYPE_FLOAT && dictionary_column(fused->columns[c], &width);
      if (!lean) break;
      uint32_t slot = FS_NARROW;
      if (width == 1) {
        lean = n_narrow < FS_NARROW;
        slot = n_narrow++;
      } else {
        lean = small_plan.column_of_slot[FS_NARROW] == 0xFFFFFFFFu;
        for (uint32_t k = 0; k < shape->n_chunks && lean; ++k) lean = reinterpret_cast<uintptr_t>(fused->columns[c]->host_segments[k].aux) % 16 == 0;   // (staged with 16-byte loads)
      }
      if (lean) { small_plan.column_of_slot[slot] = c; small_plan.slot_of_column[c] = slot; }
    }
    uint32_t n_literals = 0;
    for (uint32_t d = 0; d < n_device && lean; ++d) {
      const FusedInput& input = plan.inputs[d];
      const uint32_t function = a.aggregates[d].function;
      lean = function == HY_AGG_SUM || function == HY_AGG_AVG || function == HY_AGG_COUNT;
      if (input.n_nodes == 0) continue;   // COUNT(*): behind the accumulators with an expression
      lean = lean && d < FS_INPUTS && input.type == HY_TYPE_FLOAT;
      for (uint32_t n = 0; n < input.n_nodes && lean; ++n) {
        const FusedNode& node = input.nodes[n];
        lean = node.type == HY_TYPE_FLOAT && (node.kind == HY_EXPR_COLUMN || node.kind == HY_EXPR_LITERAL || (node.kind == HY_EXPR_ARITHMETIC && node.op <= HY_ARITH_MUL));
      }
      small_plan.n_inputs = d + 1;
      // the program: five bits per node, literals by their index in a table of FS_LITERALS.  An input that begins with the whole of the
      // input before it (Q1: l_extendedprice, then l_extendedprice * (1 - l_discount), then that * (1 + l_tax)) continues on its stack.
      uint32_t skipped = 0;
      if (d > 0 && plan.inputs[d - 1].n_nodes > 0 && plan.inputs[d - 1].n_nodes < input.n_nodes &&
          std::memcmp(plan.inputs[d - 1].nodes, input.nodes, sizeof(FusedNode) * plan.inputs[d - 1].n_nodes) == 0 && FIXED_FUSED_SHARED_PREFIX) {
        skipped = plan.inputs[d - 1].n_nodes;
      }
      for (uint32_t n = skipped; n < input.n_nodes && lean; ++n) {
        const FusedNode& node = input.nodes[n];
        uint64_t code = 0;
        if (node.kind == HY_EXPR_COLUMN) {
          lean = node.column < fused->n_columns;
          code = FS_PUSH_COLUMN + small_plan.slot_of_column[lean ? node.column : 0];
        } else if (node.kind == HY_EXPR_LITERAL) {
          uint32_t index = 0;
          while (index < n_literals && small_plan.literal[index] != static_cast<uint32_t>(node.literal)) ++index;
          if (index == n_literals) {
            lean = n_literals < FS_LITERALS;
            if (lean) small_plan.literal[n_literals++] = static_cast<uint32_t>(node.literal);
          }
          code = FS_PUSH_LITERAL + index;
        } else {
          code = node.op == HY_ARITH_ADD ? FS_ADD : node.op == HY_ARITH_SUB ? FS_SUB : FS_MUL;
        }
        small_plan.program[d] |= code << (5 * (n - skipped));
      }
      small_plan.n_nodes |= (input.n_nodes - skipped) << (4 * d);
    }
    HY_TRY(device_groups(a, shape, main_groups, reinterpret_cast<const FusedPlan*>(base), nullptr, lean ? &small_plan : nullptr));
  } else {
    // The TPC-H Q1 shape -- a handful of groups over dictionary columns, SUM / AVG / COUNT / MIN / MAX over dictionary-encoded numeric
    // columns with 1- or 2-byte value ids -- has a kernel of its own (aggregate_small.hpp); everything else takes aggregate_rows.
    SmallDomainPlan small;
    std::memset(&small, 0, sizeof(small));
    bool lean = option(HY_OPT_AGG_SMALL_DOMAIN) && shape->rows > 0 && few_codes();
    for (uint32_t k = 0; k < shape->n_chunks && lean; ++k) lean = shape->host_segments[k].size <= SD_MAX_CHUNK_ROWS;   // (a chunk is one pass of either kernel)
    std::vector<const hy_column*> inputs;   // distinct input columns, 1-byte ids first
    for (int pass = 0; pass < 2 && lean; ++pass) {
      for (uint32_t d = 0; d < n_device && lean; ++d) {
        const hy_column* column = specs[spec_of_device[d]].column;
        const uint32_t function = a.aggregates[d].function;
        lean = function == HY_AGG_SUM || function == HY_AGG_AVG || function == HY_AGG_COUNT || function == HY_AGG_MIN || function == HY_AGG_MAX;
        if (!column || !lean) continue;
        uint32_t width = 0;
        lean = (column->data_type == HY_TYPE_INT || column->data_type == HY_TYPE_LONG || column->data_type == HY_TYPE_FLOAT || column->data_type == HY_TYPE_DOUBLE) && dictionary_column(column, &width);
        if (lean && width == (pass == 0 ? 1u : 2u) && std::find(inputs.begin(), inputs.end(), column) == inputs.end()) inputs.push_back(column);
      }
      if (pass == 0) small.n_narrow = static_cast<uint32_t>(inputs.size());
    }
    lean = lean && small.n_narrow <= SD_NARROW && inputs.size() - small.n_narrow <= SD_WIDE;
    if (lean) {
      if (const char* debug = HY_DEBUG_ENV("HY_AGG_SMALL_DEBUG")) small.debug = static_cast<uint32_t>(atoi(debug));   // timing experiments only
      small.n_columns = static_cast<uint32_t>(inputs.size());
      small.joint = small.n_narrow == 2 && FIXED_AGG_JOINT_HISTOGRAM ? 1u : 0u;
      for (uint32_t k = 0; k < shape->n_chunks && small.joint; ++k) {
        if ((uint64_t{inputs[0]->host_segments[k].aux_size} + 1) * (uint64_t{inputs[1]->host_segments[k].aux_size} + 1) > SD_JOINT_CELLS) small.joint = 0;
      }
      for (uint32_t c = 0; c < small.n_columns; ++c) small.column[c] = inputs[c]->d_segments;
      for (uint32_t d = 0; d < n_device; ++d) {
        const hy_column* column = specs[spec_of_device[d]].column;
        small.column_of_aggregate[d] = column ? static_cast<uint32_t>(std::find(inputs.begin(), inputs.end(), column) - inputs.begin()) : 0xFFFFFFFFu;
        if (column && (a.aggregates[d].function == HY_AGG_MIN || a.aggregates[d].function == HY_AGG_MAX)) small.extremes |= 1u << small.column_of_aggregate[d];
      }
    }
    a.pos_cache = reads_references ? 1u : 0u;
    HY_TRY(device_groups(a, shape, main_groups, nullptr, lean ? &small : nullptr));
  }
  lap("device groups on host");
  if (main_groups.on_device) {
    const uint32_t n_groups = main_groups.n_groups;
    result->n_groups = n_groups;
    if (n_groups > result->group_capacity) return fail(HY_ERR_CAPACITY, "aggregate produces %u groups, capacity is %u", n_groups, result->group_capacity);
    const bool int_key = n_groupby == 1 && groupby[0]->data_type == HY_TYPE_INT;
    GroupExtent* extent_host = nullptr;
    GroupExtent* extent_dev = nullptr;
    HY_TRY(pinned_staging(sizeof(GroupExtent), reinterpret_cast<void**>(&extent_host), reinterpret_cast<void**>(&extent_dev)));
    const uint32_t blocks = (n_groups + 255) / 256;
    DeviceBuffer extent_partial;
    HY_TRY(extent_partial.alloc(sizeof(GroupExtent) + 8));
    HY_HIP(hipMemsetAsync(extent_partial.ptr, 0, sizeof(GroupExtent) + 8, stream));
    HY_HIP(hipMemsetAsync(extent_partial.ptr, 0xFF, 8, stream));   // key_min
    hipLaunchKernelGGL(finish_extent, dim3(std::min(blocks, 1024u)), dim3(256), 0, stream, main_groups.d_keys.as<uint64_t>(), main_groups.d_first.as<uint64_t>(), n_groups, words, int_key ? 1u : 0u,
                       extent_partial.as<GroupExtent>(), extent_dev);
    HY_HIP(hipStreamSynchronize(stream));
    const uint64_t min_key = extent_host->key_min, max_key = extent_host->key_max;
    // the immediate-key shortcut, as below (aggregate_hash.cpp:388-401)
    const bool immediate = int_key && max_key > 0 && static_cast<double>(max_key - min_key) < static_cast<double>(fused ? main_groups.passed_rows : shape->rows) * 1.2;
    const uint64_t largest_sort_key = immediate ? max_key - min_key + 1 : extent_host->row_max;
    if (largest_sort_key >= (1ull << 32)) HY_TRY(download_groups(main_groups, words, n_device, stream));   // (keys spread over more than 2^32 values with more rows than that to justify it: never)
    else {
      uint32_t key_bits = 1;
      while (key_bits < 32 && (largest_sort_key >> key_bits) != 0) ++key_bits;
      DeviceBuffer sort_keys, sort_ids, sort_keys_tmp, sort_ids_tmp;
      HY_TRY(sort_keys.alloc(4 * size_t{n_groups}));
      HY_TRY(sort_ids.alloc(4 * size_t{n_groups}));
      HY_TRY(sort_keys_tmp.alloc(4 * size_t{n_groups}));
      HY_TRY(sort_ids_tmp.alloc(4 * size_t{n_groups}));
      hipLaunchKernelGGL(finish_sort_keys, dim3(blocks), dim3(256), 0, stream, main_groups.d_keys.as<uint64_t>(), main_groups.d_first.as<uint64_t>(), n_groups, words, immediate ? 1u : 0u, min_key,
                         sort_keys.as<uint32_t>(), sort_ids.as<uint32_t>());
      uint32_t* sorted_keys = sort_keys.as<uint32_t>();
      uint32_t* order = sort_ids.as<uint32_t>();
      HY_TRY(sort_pairs_u32(&sorted_keys, &order, sort_keys_tmp.as<uint32_t>(), sort_ids_tmp.as<uint32_t>(), n_groups, key_bits, stream));
      // Results up to 16 MiB are written straight into pinned host memory by the kernels and copied out by the host (every hipMemcpyAsync
      // into pageable memory is a staged copy with ~0.2 ms of host work around it: five of them cost more than the kernels); larger ones
      // go through device arrays and one copy each.
      const RowIdOf row_id_of(shape);
      size_t value_bytes[MAX_AGGREGATES], values_at[MAX_AGGREGATES], nulls_at[MAX_AGGREGATES];
      size_t staged_bytes = result->group_row_ids ? align_up(sizeof(hy_row_id) * size_t{n_groups}, 256) : 0;
      for (uint32_t g = 0; g < n_aggregates; ++g) {
        hy_aggregate_column& col = result->columns[g];
        col.data_type = result_type(specs[g].function, input_type(g));
        value_bytes[g] = (col.data_type == HY_TYPE_INT || col.data_type == HY_TYPE_FLOAT) ? 4 : 8;
        values_at[g] = staged_bytes;
        staged_bytes += align_up(value_bytes[g] * n_groups, 256);
        nulls_at[g] = staged_bytes;
        if (col.is_null) staged_bytes += align_up(size_t{n_groups}, 256);
      }
      const bool through_pinned = staged_bytes <= (size_t{16} << 20);
      unsigned char* staged_host = nullptr;
      unsigned char* staged_dev = nullptr;
      DeviceBuffer staged_device_memory;
      if (through_pinned) HY_TRY(pinned_staging(staged_bytes, reinterpret_cast<void**>(&staged_host), reinterpret_cast<void**>(&staged_dev)));
      else {
        HY_TRY(staged_device_memory.alloc(staged_bytes));
        staged_dev = staged_device_memory.as<unsigned char>();
      }
      if (result->group_row_ids) {
        hipLaunchKernelGGL(finish_row_ids, dim3(blocks), dim3(256), 0, stream, order, immediate ? main_groups.d_last.as<uint64_t>() : main_groups.d_first.as<uint64_t>(), n_groups, shape->d_row_base,
                           shape->n_chunks, static_cast<uint32_t>(row_id_of.size), reinterpret_cast<hy_row_id*>(staged_dev));
        if (!through_pinned) HY_HIP(hipMemcpyAsync(result->group_row_ids, staged_dev, sizeof(hy_row_id) * size_t{n_groups}, hipMemcpyDeviceToHost, stream));
      }
      for (uint32_t g = 0; g < n_aggregates; ++g) {
        hy_aggregate_column& col = result->columns[g];
        const uint32_t in_type = input_type(g);
        const FinishColumn c{specs[g].function, col.data_type, (in_type == HY_TYPE_FLOAT || in_type == HY_TYPE_DOUBLE) ? 1u : 0u, static_cast<uint32_t>(primary[g])};
        hipLaunchKernelGGL(finish_column, dim3(blocks), dim3(256), 0, stream, order, main_groups.d_values.as<uint64_t>(), main_groups.d_counts.as<uint64_t>(), n_groups, n_device, c, staged_dev + values_at[g],
                           col.is_null ? staged_dev + nulls_at[g] : nullptr);
        if (through_pinned) continue;
        HY_HIP(hipMemcpyAsync(col.values, staged_dev + values_at[g], value_bytes[g] * n_groups, hipMemcpyDeviceToHost, stream));
        if (col.is_null) HY_HIP(hipMemcpyAsync(col.is_null, staged_dev + nulls_at[g], n_groups, hipMemcpyDeviceToHost, stream));
      }
      HY_HIP(hipStreamSynchronize(stream));
      if (through_pinned) {
        if (result->group_row_ids) std::memcpy(result->group_row_ids, staged_host, sizeof(hy_row_id) * size_t{n_groups});
        for (uint32_t g = 0; g < n_aggregates; ++g) {
          std::memcpy(result->columns[g].values, staged_host + values_at[g], value_bytes[g] * n_groups);
          if (result->columns[g].is_null) std::memcpy(result->columns[g].is_null, staged_host + nulls_at[g], n_groups);
        }
      }
      g_agg_finished_on_device = 1;
      lap("finished on the device");
      return HY_OK;
    }
  }
  g_agg_finished_on_device = 0;
  const uint32_t n_groups = main_groups.n_groups;
  const std::vector<uint64_t>&h_keys = main_groups.keys, &h_first = main_groups.first, &h_last = main_groups.last, &h_values = main_groups.values,
                             &h_counts = main_groups.counts;

  // COUNT(DISTINCT column): groups of (GROUP BY columns, column) with a non-NULL column value, counted per outer group
  std::vector<std::vector<uint64_t>> distinct_counts(n_aggregates);
  for (uint32_t g = 0; g < n_aggregates; ++g) {
    if (specs[g].function != HY_AGG_COUNT_DISTINCT) continue;
    AggArgs inner = a;
    inner.n_groupby = n_groupby + 1;
    inner.n_aggregates = 0;
    wire(inner.groupby[n_groupby], specs[g].column, 0);
    inner.pos_cache = reads_references ? 1u : 0u;
    DeviceGroups combos;
    HY_TRY(device_groups(inner, shape, combos));
    std::map<std::vector<uint64_t>, uint32_t> group_of_key;
    for (uint32_t i = 0; i < n_groups; ++i) group_of_key.emplace(std::vector<uint64_t>(h_keys.begin() + size_t{i} * words, h_keys.begin() + size_t{i + 1} * words), i);
    distinct_counts[g].assign(n_groups, 0);
    const uint32_t inner_words = words + 1;
    for (uint32_t i = 0; i < combos.n_groups; ++i) {
      const uint64_t* tuple = combos.keys.data() + size_t{i} * inner_words;
      if ((tuple[0] >> n_groupby) & 1) continue;   // the aggregate column is NULL in this combination
      std::vector<uint64_t> outer(tuple, tuple + words);
      outer[0] &= (1ull << n_groupby) - 1;
      const auto it = group_of_key.find(outer);
      if (it != group_of_key.end()) distinct_counts[g][it->second] += 1;
    }
  }

  // ---- order of the result rows (aggregate_hash.cpp:388-401, 770-804) ---------------------------------------------------
  std::vector<uint32_t> order(n_groups);
  for (uint32_t i = 0; i < n_groups; ++i) order[i] = i;
  bool immediate = false;
  if (n_groupby == 1 && groupby[0]->data_type == HY_TYPE_INT) {
    uint64_t min_key = ~0ull, max_key = 0;
    for (uint32_t i = 0; i < n_groups; ++i) {
      if (h_keys[size_t{i} * words] & 1) continue;   // NULL group
      const uint64_t k = static_cast<uint64_t>(static_cast<int64_t>(h_keys[size_t{i} * words + 1]) - static_cast<int64_t>(INT32_MIN)) + 1;
      min_key = std::min(min_key, k);
      max_key = std::max(max_key, k);
    }
    // (the row count is that of the aggregate's INPUT: behind fused filters, the rows that passed them)
    immediate = max_key > 0 && static_cast<double>(max_key - min_key) < static_cast<double>(fused ? main_groups.passed_rows : shape->rows) * 1.2;
  }
  {   // ascending key (NULL first) or first row: an LSD radix sort of (64-bit sort key, group) -- a comparison sort of 100 000 groups
      // through an index costs more than the device spends on the whole table
    std::vector<uint64_t> sort_key(n_groups);
    uint64_t all_bits = 0;
    for (uint32_t i = 0; i < n_groups; ++i) {
      if (immediate) sort_key[i] = (h_keys[size_t{i} * words] & 1) ? 0 : static_cast<uint64_t>(static_cast<int64_t>(h_keys[size_t{i} * words + 1]) - static_cast<int64_t>(INT32_MIN)) + 1;
      else sort_key[i] = h_first[i];
      all_bits |= sort_key[i];
    }
    std::vector<uint32_t> scratch_order(n_groups);
    if (n_groups <= 2048) {   // (a handful of groups -- TPC-H Q1 has four: two passes over 65 536 buckets cost 17 us of a 0.28 ms call)
      std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return sort_key[x] < sort_key[y]; });
      all_bits = 0;
    }
    for (uint32_t shift = 0; shift < 64 && (all_bits >> shift) != 0; shift += 16) {
      std::vector<uint32_t> bucket(65537, 0);
      for (uint32_t i = 0; i < n_groups; ++i) ++bucket[((sort_key[order[i]] >> shift) & 0xFFFF) + 1];
      for (uint32_t b = 0; b < 65536; ++b) bucket[b + 1] += bucket[b];
      for (uint32_t i = 0; i < n_groups; ++i) scratch_order[bucket[(sort_key[order[i]] >> shift) & 0xFFFF]++] = order[i];
      order.swap(scratch_order);
    }
  }

  lap("groups ordered");
  const bool no_groupby_empty = n_groupby == 0 && n_groups == 0;   // one row of NULLs / zero counts (:1422-1432)
  const uint32_t out_groups = no_groupby_empty ? 1 : n_groups;
  result->n_groups = out_groups;
  if (out_groups > result->group_capacity) return fail(HY_ERR_CAPACITY, "aggregate produces %u groups, capacity is %u", out_groups, result->group_capacity);
  if (result->mem != HY_MEM_HOST) return fail(HY_ERR_UNSUPPORTED, "aggregate results are returned in host memory (they are ordered on the host)");
  std::vector<hy_row_id> representative(out_groups, hy_row_id{0, 0});
  const RowIdOf row_id_of(shape);
  for_each_group_range(n_groups, [&](uint32_t begin, uint32_t end) {
    const uint64_t* rows_of_groups = immediate ? h_last.data() : h_first.data();
    for (uint32_t o = begin; o < end; ++o) {
      if (o + 16 < end) __builtin_prefetch(rows_of_groups + order[o + 16]);   // (the groups come in table order, the result in its own)
      representative[o] = row_id_of(rows_of_groups[order[o]]);
    }
  });
  if (result->group_row_ids) std::memcpy(result->group_row_ids, representative.data(), sizeof(hy_row_id) * out_groups);
  lap("representatives");

  for (uint32_t g = 0; g < n_aggregates; ++g) {
    hy_aggregate_column& col = result->columns[g];
    const uint32_t function = specs[g].function;
    const uint32_t in_type = input_type(g);
    const bool is_float = in_type == HY_TYPE_FLOAT || in_type == HY_TYPE_DOUBLE;
    col.data_type = result_type(function, in_type);
    if (!col.values) return fail(HY_ERR_INVALID, "aggregate %u: values buffer missing", g);
    std::vector<uint64_t> any_bits;
    std::vector<uint8_t> any_null;
    if (function == HY_AGG_ANY && n_groups) {
      DeviceBuffer d_rows, d_bits, d_null;
      HY_TRY(d_rows.alloc(8 * size_t{n_groups}));
      HY_TRY(d_bits.alloc(8 * size_t{n_groups}));
      HY_TRY(d_null.alloc(n_groups));
      HY_HIP(hipMemcpyAsync(d_rows.ptr, representative.data(), 8 * size_t{n_groups}, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(gather_values, dim3((n_groups + 255) / 256), dim3(256), 0, stream, specs[g].column->d_segments, d_rows.as<hy_row_id>(), n_groups, is_float ? 1u : 0u,
                         d_bits.as<uint64_t>(), d_null.as<uint8_t>());
      any_bits.resize(n_groups);
      any_null.resize(n_groups);
      HY_HIP(hipMemcpyAsync(any_bits.data(), d_bits.ptr, 8 * size_t{n_groups}, hipMemcpyDeviceToHost, stream));
      HY_HIP(hipMemcpyAsync(any_null.data(), d_null.ptr, n_groups, hipMemcpyDeviceToHost, stream));
      HY_HIP(hipStreamSynchronize(stream));
    }
    // the common shapes as tight loops (a result of 100 000 groups spends more time here than the device spends on the table):
    // COUNT -> int64, SUM / AVG of a floating-point column or AVG of an integer one -> double, SUM of an integer column -> int64
    const bool plain = out_groups == n_groups && primary[g] >= 0;
    const uint64_t* group_values = h_values.data() + (primary[g] >= 0 ? primary[g] : 0);
    const uint64_t* group_counts = h_counts.data() + (primary[g] >= 0 ? primary[g] : 0);
    if (plain && function == HY_AGG_COUNT && col.data_type == HY_TYPE_LONG) {
      for_each_group_range(out_groups, [&](uint32_t range_begin, uint32_t range_end) {
        int64_t* out_values = static_cast<int64_t*>(col.values);
        for (uint32_t o = range_begin; o < range_end; ++o) {
          if (o + 16 < range_end) __builtin_prefetch(group_counts + size_t{order[o + 16]} * n_device);
          out_values[o] = static_cast<int64_t>(group_counts[size_t{order[o]} * n_device]);
        }
        if (col.is_null) std::memset(col.is_null + range_begin, 0, range_end - range_begin);
      });
      continue;
    }
    if (plain && (function == HY_AGG_SUM || function == HY_AGG_AVG) && (col.data_type == HY_TYPE_DOUBLE || (function == HY_AGG_SUM && col.data_type == HY_TYPE_LONG))) {
      const bool as_double = col.data_type == HY_TYPE_DOUBLE, divide = function == HY_AGG_AVG;
      for_each_group_range(out_groups, [&](uint32_t range_begin, uint32_t range_end) {
        for (uint32_t o = range_begin; o < range_end; ++o) {
          if (o + 16 < range_end) {
            __builtin_prefetch(group_values + size_t{order[o + 16]} * n_device);
            __builtin_prefetch(group_counts + size_t{order[o + 16]} * n_device);
          }
          const size_t at = size_t{order[o]} * n_device;
          const uint64_t bits = group_values[at], count = group_counts[at];
          const bool is_null = count == 0;
          if (col.is_null) col.is_null[o] = is_null;
          if (as_double) {
            double sum;
            std::memcpy(&sum, &bits, 8);
            static_cast<double*>(col.values)[o] = is_null ? 0.0 : (divide ? sum / static_cast<double>(count) : sum);
          } else {
            static_cast<int64_t*>(col.values)[o] = is_null ? 0 : static_cast<int64_t>(bits);
          }
        }
      });
      continue;
    }
    for_each_group_range(out_groups, [&](uint32_t range_begin, uint32_t range_end) {
    for (uint32_t o = range_begin; o < range_end; ++o) {
      const bool have = o < n_groups;
      const uint64_t bits = have && primary[g] >= 0 ? h_values[size_t{order[o]} * n_device + primary[g]] : 0;
      const uint64_t count = have && primary[g] >= 0 ? h_counts[size_t{order[o]} * n_device + primary[g]] : 0;
      bool is_null = false;
      int64_t vi = 0;
      double vf = 0.0;
      switch (function) {
        case HY_AGG_COUNT: vi = static_cast<int64_t>(count); break;
        case HY_AGG_COUNT_DISTINCT: vi = have ? static_cast<int64_t>(distinct_counts[g][order[o]]) : 0; break;
        case HY_AGG_STDDEV_SAMP:   // abstract_aggregate_operator.hpp:83-113 (Welford there; sums of the values shifted by a value of the column here)
          is_null = count <= 1;
          if (count > 1) {
            double sum, squares;
            std::memcpy(&sum, &bits, 8);
            std::memcpy(&squares, &h_values[size_t{order[o]} * n_device + secondary[g]], 8);
            const double n = static_cast<double>(count);
            const double variance = (squares - sum * sum / n) / (n - 1.0);
            vf = variance > 0.0 ? std::sqrt(variance) : 0.0;
          }
          break;
        case HY_AGG_SUM:
          is_null = count == 0;
          if (is_float) std::memcpy(&vf, &bits, 8); else vi = static_cast<int64_t>(bits);
          break;
        case HY_AGG_AVG:
          is_null = count == 0;
          if (count) { double sum; std::memcpy(&sum, &bits, 8); vf = sum / static_cast<double>(count); }   // aggregate_hash.cpp:166
          break;
        case HY_AGG_MIN:
        case HY_AGG_MAX:
          is_null = count == 0;
          if (is_float) {
            int64_t ordered = static_cast<int64_t>(bits);
            if (ordered < 0) ordered ^= 0x7FFFFFFFFFFFFFFFll;
            std::memcpy(&vf, &ordered, 8);
          } else vi = static_cast<int64_t>(bits);
          break;
        default:   // ANY
          is_null = !have || any_null[o];
          if (have && !is_null) { if (is_float) std::memcpy(&vf, &any_bits[o], 8); else vi = static_cast<int64_t>(any_bits[o]); }
          break;
      }
      if (col.is_null) col.is_null[o] = is_null;
      switch (col.data_type) {
        case HY_TYPE_INT: static_cast<int32_t*>(col.values)[o] = is_null ? 0 : static_cast<int32_t>(vi); break;
        case HY_TYPE_LONG: static_cast<int64_t*>(col.values)[o] = is_null ? 0 : vi; break;
        case HY_TYPE_FLOAT: static_cast<float*>(col.values)[o] = is_null ? 0.f : static_cast<float>(vf); break;
        default: static_cast<double*>(col.values)[o] = is_null ? 0.0 : vf; break;
      }
    }
    });
  }
  lap("result written");
  return HY_OK;
}

// hy_scan_project_aggregate: checks the plan, types the expressions (expression_common_type per arithmetic node, like
// hy_projection_arithmetic per call) and runs the aggregate with fused_rows in the place of aggregate_rows.
static hy_status run_fused(const hy_filter* filters, uint32_t n_filters, const hy_column* const* groupby, uint32_t n_groupby, const hy_fused_aggregate* aggregates,
                           uint32_t n_aggregates, hy_aggregate_result* result) {
  if (n_filters > HY_MAX_FILTERS) return fail(HY_ERR_UNSUPPORTED, "more than %d fused filters: run the operator chain", static_cast<int>(HY_MAX_FILTERS));
  if (n_aggregates > MAX_AGGREGATES) return fail(HY_ERR_UNSUPPORTED, "more than %u aggregates stay on the CPU path", MAX_AGGREGATES);
  if (n_groupby > MAX_GROUPBY) return fail(HY_ERR_UNSUPPORTED, "more than %u GROUP BY columns stay on the CPU path", MAX_GROUPBY);
  auto query = std::make_unique<FusedQuery>();
  FusedQuery& q = *query;
  std::memset(&q, 0, sizeof(q));
  const hy_column* shape = nullptr;
  auto table_column = [&](const hy_column* c, const char* what) -> hy_status {
    if (!c) return fail(HY_ERR_INVALID, "hy_scan_project_aggregate: %s column missing", what);
    if (c->is_reference || c->is_mvcc) return fail(HY_ERR_UNSUPPORTED, "hy_scan_project_aggregate reads the columns of a data table (%s column is a reference / MVCC column): run the operator chain", what);
    if (!shape) { shape = c; return HY_OK; }
    if (c->n_chunks != shape->n_chunks) return fail(HY_ERR_INVALID, "%s column does not belong to the table (chunk counts differ)", what);
    for (uint32_t k = 0; k < c->n_chunks; ++k) {
      if (c->host_segments[k].size != shape->host_segments[k].size) return fail(HY_ERR_INVALID, "%s column does not belong to the table (chunk %u)", what, k);
    }
    return HY_OK;
  };
  auto numeric = [](uint32_t t) { return t >= HY_TYPE_INT && t <= HY_TYPE_DOUBLE; };
  for (uint32_t f = 0; f < n_filters; ++f) {
    if (filters[f].predicate.condition == HY_FILTER_VALIDATE) {   // the table's MvccData: same chunk layout, checked like a column
      const hy_column* mvcc = filters[f].column;
      if (!mvcc || !mvcc->is_mvcc || mvcc->is_reference) return fail(HY_ERR_INVALID, "a Validate filter needs the table's column of HY_ENC_MVCC segments");
      if (shape && mvcc->n_chunks != shape->n_chunks) return fail(HY_ERR_INVALID, "the MvccData does not belong to the table (chunk counts differ)");
      if (!shape) shape = mvcc;
      continue;
    }
    HY_TRY(table_column(filters[f].column, "filter"));
  }
  for (uint32_t g = 0; g < n_groupby; ++g) HY_TRY(table_column(groupby[g], "GROUP BY"));
  for (uint32_t g = 0; g < n_aggregates; ++g) {
    const hy_expression* e = aggregates[g].input;
    FusedInput& input = q.inputs[g];
    q.same_as[g] = g;
    if (!e) continue;   // COUNT(*)
    if (e->n_nodes == 0 || e->n_nodes > HY_MAX_EXPRESSION_NODES) return fail(HY_ERR_INVALID, "aggregate %u: an expression has 1 to %d nodes", g, static_cast<int>(HY_MAX_EXPRESSION_NODES));
    uint32_t types[4] = {0, 0, 0, 0}, depth = 0;
    int producers[4] = {-1, -1, -1, -1};   // the LITERAL node that put the stack entry there, else -1
    for (uint32_t k = 0; k < e->n_nodes; ++k) {
      const hy_expression_node& n = e->nodes[k];
      FusedNode& out = input.nodes[k];
      out.kind = n.kind;
      if (n.kind == HY_EXPR_ARITHMETIC) {
        if (n.op > HY_ARITH_MOD) return fail(HY_ERR_INVALID, "aggregate %u: unknown arithmetic operator %u", g, n.op);
        if (depth < 2) return fail(HY_ERR_INVALID, "aggregate %u: node %u has no two operands below it (postfix order)", g, k);
        const uint32_t left = types[depth - 2], right = types[depth - 1];
        if (left == HY_TYPE_NULL && right == HY_TYPE_NULL) return fail(HY_ERR_INVALID, "Cannot deduce common type if both sides are NULL.");
        out.op = n.op;
        out.type = expression_common_type(left, right);
        // A literal operand of + - * is converted to the operation's C++ type before anything is computed (arithmetic_cell): done here, once,
        // so that the kernel finds two operands of the result's own type and takes its one-instruction path.
        if (n.op <= HY_ARITH_MUL) {
          for (int side = 0; side < 2; ++side) {
            const uint32_t at = side == 0 ? left : right, other = side == 0 ? right : left;
            const int producer = side == 0 ? producers[depth - 2] : producers[depth - 1];
            if (producer < 0 || at == HY_TYPE_NULL || other == HY_TYPE_NULL) continue;
            FusedNode& literal = input.nodes[producer];
            const uint32_t common = (at == HY_TYPE_DOUBLE || other == HY_TYPE_DOUBLE) ? HY_TYPE_DOUBLE : (at == HY_TYPE_FLOAT || other == HY_TYPE_FLOAT) ? HY_TYPE_FLOAT
                                    : (at == HY_TYPE_LONG || other == HY_TYPE_LONG) ? HY_TYPE_LONG : HY_TYPE_INT;   // std::common_type_t
            if (common == at || common != out.type) continue;   // (int (op) long literal: the result type is the common type as well; long (op) float -> double is not)
            double as_double = 0.0;
            float as_float = 0.f;
            int64_t as_integer = static_cast<int64_t>(literal.literal);
            if (at == HY_TYPE_FLOAT) { uint32_t word = static_cast<uint32_t>(literal.literal); std::memcpy(&as_float, &word, 4); as_double = as_float; }
            else as_double = static_cast<double>(as_integer);
            if (common == HY_TYPE_DOUBLE) std::memcpy(&literal.literal, &as_double, 8);
            else if (common == HY_TYPE_FLOAT) { as_float = static_cast<float>(as_double); uint32_t word; std::memcpy(&word, &as_float, 4); literal.literal = word; }   // (via double, like the device's convert())
            else literal.literal = static_cast<uint64_t>(as_integer);   // int -> long
            literal.type = common;
          }
        }
        depth -= 1;
        types[depth - 1] = out.type;
        producers[depth - 1] = -1;
        continue;
      }
      if (depth == 3) return fail(HY_ERR_UNSUPPORTED, "aggregate %u: the expression needs more than three stack slots: run the operator chain", g);
      if (n.kind == HY_EXPR_COLUMN) {
        HY_TRY(table_column(n.column, "expression"));
        if (!numeric(n.column->data_type)) return fail(HY_ERR_UNSUPPORTED, "string expressions stay on the CPU path");
        uint32_t which = 0;
        while (which < q.n_columns && q.columns[which] != n.column) ++which;
        if (which == q.n_columns) {
          if (q.n_columns == static_cast<uint32_t>(FUSED_COLUMNS)) return fail(HY_ERR_UNSUPPORTED, "the expressions read more than %d distinct columns: run the operator chain", FUSED_COLUMNS);
          q.columns[q.n_columns++] = n.column;
        }
        out.column = which;
        out.type = n.column->data_type;
      } else if (n.kind == HY_EXPR_LITERAL) {
        if (n.literal_type != HY_TYPE_NULL && !numeric(n.literal_type)) return fail(HY_ERR_UNSUPPORTED, "string expressions stay on the CPU path");
        out.type = n.literal_type;
        double widened = 0.0;
        switch (n.literal_type) {
          case HY_TYPE_INT: out.literal = static_cast<uint64_t>(static_cast<int64_t>(n.literal.i32)); break;
          case HY_TYPE_LONG: out.literal = static_cast<uint64_t>(n.literal.i64); break;
          case HY_TYPE_FLOAT: { uint32_t word; std::memcpy(&word, &n.literal.f32, 4); out.literal = word; break; }   // (floats travel as their own bits)
          case HY_TYPE_DOUBLE: widened = n.literal.f64; std::memcpy(&out.literal, &widened, 8); break;
          default: break;
        }
      } else {
        return fail(HY_ERR_INVALID, "aggregate %u: unknown expression node kind %u", g, n.kind);
      }
      producers[depth] = n.kind == HY_EXPR_LITERAL ? static_cast<int>(k) : -1;
      types[depth++] = out.type;
    }
    if (depth != 1) return fail(HY_ERR_INVALID, "aggregate %u: the expression leaves %u results (postfix order)", g, depth);
    if (types[0] == HY_TYPE_NULL) return fail(HY_ERR_UNSUPPORTED, "aggregate %u: an aggregate of the NULL literal stays on the CPU path", g);
    input.n_nodes = e->n_nodes;
    input.type = types[0];
    for (uint32_t earlier = 0; earlier < g; ++earlier) {
      if (std::memcmp(&q.inputs[earlier], &input, sizeof(FusedInput)) == 0) { q.same_as[g] = q.same_as[earlier]; break; }
    }
  }
  if (!shape) return fail(HY_ERR_INVALID, "hy_scan_project_aggregate needs at least one column");
  q.shape = shape;
  q.filters = filters;
  q.n_filters = n_filters;
  std::vector<hy_aggregate_spec> specs(n_aggregates ? n_aggregates : 1);
  for (uint32_t g = 0; g < n_aggregates; ++g) specs[g] = hy_aggregate_spec{aggregates[g].function, nullptr};
  return run_aggregate(groupby, n_groupby, specs.data(), n_aggregates, result, &q);
}

}  // namespace HY_AGG_NAMESPACE
}  // namespace hy

using namespace hy;

// HY_MEM_DEVICE results: the groups are ordered on the host (the reference's order is a host-side sort of a few groups), so the
// operator runs against a host shadow of the caller's buffers and the filled prefix is uploaded -- what the caller gets is a result a
// next operator can read without the values crossing to the host's address space of the CALLER (an operator chain that ends on the device).
template <typename Run>
static hy_status run_with_result_memory(hy_aggregate_result* result, uint32_t n_aggregates, Run run) {
  if (result->mem == HY_MEM_HOST) return run(result);
  if (result->mem != HY_MEM_DEVICE) return fail(HY_ERR_INVALID, "aggregate result: bad memory space %u", result->mem);
  const size_t capacity = result->group_capacity;
  std::vector<hy_row_id> rows(capacity ? capacity : 1);
  std::vector<std::vector<uint64_t>> values(n_aggregates, std::vector<uint64_t>(capacity ? capacity : 1));
  std::vector<std::vector<uint8_t>> nulls(n_aggregates, std::vector<uint8_t>(capacity ? capacity : 1));
  std::vector<hy_aggregate_column> columns(n_aggregates ? n_aggregates : 1);
  for (uint32_t a = 0; a < n_aggregates; ++a) { columns[a].values = values[a].data(); columns[a].is_null = result->columns[a].is_null ? nulls[a].data() : nullptr; }
  hy_aggregate_result shadow = *result;
  shadow.mem = HY_MEM_HOST;
  shadow.group_row_ids = result->group_row_ids ? rows.data() : nullptr;
  shadow.columns = columns.data();
  const hy_status status = run(&shadow);
  result->n_groups = shadow.n_groups;
  if (status != HY_OK) return status;
  hipStream_t stream = current_stream();
  const size_t n = shadow.n_groups;
  if (result->group_row_ids && n) HY_HIP(hipMemcpyAsync(result->group_row_ids, rows.data(), sizeof(hy_row_id) * n, hipMemcpyHostToDevice, stream));
  for (uint32_t a = 0; a < n_aggregates; ++a) {
    result->columns[a].data_type = columns[a].data_type;
    const size_t width = columns[a].data_type == HY_TYPE_INT || columns[a].data_type == HY_TYPE_FLOAT ? 4 : 8;
    if (!result->columns[a].values) return fail(HY_ERR_INVALID, "aggregate %u: values buffer missing", a);
    if (n) HY_HIP(hipMemcpyAsync(result->columns[a].values, values[a].data(), width * n, hipMemcpyHostToDevice, stream));
    if (result->columns[a].is_null && n) HY_HIP(hipMemcpyAsync(result->columns[a].is_null, nulls[a].data(), n, hipMemcpyHostToDevice, stream));
  }
  HY_HIP(hipStreamSynchronize(stream));   // (the shadows die with this call)
  return HY_OK;
}

extern "C" {

#ifdef HY_AGG_NEXT
// aggregate_wide.hip / aggregate_widest.hip: this file again, with nine- / seventeen-word tuples
hy_status HY_AGG_NEXT(hy_scan_project_aggregate)(const hy_filter* filters, uint32_t n_filters, const hy_column* const* groupby_columns, uint32_t n_groupby,
                                                 const hy_fused_aggregate* aggregates, uint32_t n_aggregates, hy_aggregate_result* result);
hy_status HY_AGG_NEXT(hy_aggregate_hash)(const hy_column* const* groupby_columns, uint32_t n_groupby, const hy_aggregate_spec* aggregates, uint32_t n_aggregates,
                                         hy_aggregate_result* result);
int HY_AGG_NEXT(hy_debug_aggregate_path)(void);
int HY_AGG_NEXT(hy_debug_aggregate_finished_on_device)(void);
int HY_AGG_NEXT(hy_debug_aggregate_small_domain)(void);
static bool g_last_aggregate_was_wide = false;   // debug accessors: the next build answered the last call of this process
#endif

hy_status HY_AGG_ENTRY(hy_scan_project_aggregate)(const hy_filter* filters, uint32_t n_filters, const hy_column* const* groupby_columns, uint32_t n_groupby,
                                                  const hy_fused_aggregate* aggregates, uint32_t n_aggregates, hy_aggregate_result* result) {
#ifdef HY_AGG_NEXT
  g_last_aggregate_was_wide = n_groupby > MAX_GROUPBY;
  if (n_groupby > MAX_GROUPBY) return HY_AGG_NEXT(hy_scan_project_aggregate)(filters, n_filters, groupby_columns, n_groupby, aggregates, n_aggregates, result);
#endif
  if (!result || (n_filters && !filters) || (n_groupby && !groupby_columns) || (n_aggregates && !aggregates)) return fail(HY_ERR_INVALID, "hy_scan_project_aggregate: null argument");
  if (n_aggregates && !result->columns) return fail(HY_ERR_INVALID, "hy_scan_project_aggregate: result columns missing");
  // every column of the plan lives on the calling thread's device (hy_bind_device: one worker thread per GPU) -- checked before a compressed
  // column's twin is decoded, which would happen on the wrong device with this thread's pool
  for (uint32_t f = 0; f < n_filters; ++f) HY_TRY(on_this_device(filters[f].column, "hy_scan_project_aggregate"));
  for (uint32_t i = 0; i < n_groupby; ++i) HY_TRY(on_this_device(groupby_columns[i], "hy_scan_project_aggregate"));
  for (uint32_t g = 0; g < n_aggregates; ++g) {
    if (!aggregates[g].input) continue;
    for (uint32_t k = 0; k < aggregates[g].input->n_nodes && k < HY_MAX_EXPRESSION_NODES; ++k)
      if (aggregates[g].input->nodes[k].kind == HY_EXPR_COLUMN) HY_TRY(on_this_device(aggregates[g].input->nodes[k].column, "hy_scan_project_aggregate"));
  }
  // run-length / bit-packed segments: the decoded twins (hy_device.hpp)
  std::vector<hy_filter> plain_filters(filters, filters + n_filters);
  for (hy_filter& filter : plain_filters) HY_TRY(plain_column(filter.column, &filter.column));
  std::vector<const hy_column*> plain_groupby(groupby_columns, groupby_columns + n_groupby);
  for (const hy_column*& column : plain_groupby) HY_TRY(plain_column(column, &column));
  std::vector<hy_fused_aggregate> plain_aggregates(aggregates, aggregates + n_aggregates);
  std::vector<hy_expression> plain_inputs(n_aggregates);
  for (uint32_t g = 0; g < n_aggregates; ++g) {
    if (!aggregates[g].input) continue;
    plain_inputs[g] = *aggregates[g].input;
    for (uint32_t k = 0; k < plain_inputs[g].n_nodes && k < HY_MAX_EXPRESSION_NODES; ++k) {
      if (plain_inputs[g].nodes[k].kind == HY_EXPR_COLUMN) HY_TRY(plain_column(plain_inputs[g].nodes[k].column, &plain_inputs[g].nodes[k].column));
    }
    plain_aggregates[g].input = &plain_inputs[g];
  }
  return run_with_result_memory(result, n_aggregates, [&](hy_aggregate_result* out) {
    return run_fused(plain_filters.data(), n_filters, plain_groupby.data(), n_groupby, plain_aggregates.data(), n_aggregates, out);
  });
}

hy_status HY_AGG_ENTRY(hy_aggregate_hash)(const hy_column* const* groupby_columns, uint32_t n_groupby, const hy_aggregate_spec* aggregates,
                                          uint32_t n_aggregates, hy_aggregate_result* result) {
#ifdef HY_AGG_NEXT
  {   // (COUNT(DISTINCT x) groups by the GROUP BY columns and x: one more key word)
    uint32_t key_columns = n_groupby;
    for (uint32_t i = 0; aggregates && i < n_aggregates; ++i) if (aggregates[i].function == HY_AGG_COUNT_DISTINCT) key_columns = n_groupby + 1;
    g_last_aggregate_was_wide = key_columns > MAX_GROUPBY;
    if (g_last_aggregate_was_wide) return HY_AGG_NEXT(hy_aggregate_hash)(groupby_columns, n_groupby, aggregates, n_aggregates, result);
  }
#endif
  if (!result || (n_groupby && !groupby_columns) || (n_aggregates && !aggregates)) return fail(HY_ERR_INVALID, "hy_aggregate_hash: null argument");
  if (n_aggregates && !result->columns) return fail(HY_ERR_INVALID, "hy_aggregate_hash: result columns missing");
  for (uint32_t i = 0; i < n_groupby; ++i) HY_TRY(on_this_device(groupby_columns[i], "hy_aggregate_hash"));
  for (uint32_t i = 0; i < n_aggregates; ++i) HY_TRY(on_this_device(aggregates[i].column, "hy_aggregate_hash"));
  std::vector<const hy_column*> plain_groupby(groupby_columns, groupby_columns + n_groupby);   // run-length / bit-packed segments: the decoded twins
  for (const hy_column*& column : plain_groupby) HY_TRY(plain_column(column, &column));
  std::vector<hy_aggregate_spec> plain_aggregates(aggregates, aggregates + n_aggregates);
  for (hy_aggregate_spec& spec : plain_aggregates) HY_TRY(plain_column(spec.column, &spec.column));
  return run_with_result_memory(result, n_aggregates, [&](hy_aggregate_result* out) -> hy_status {
    // More aggregates than one pass has accumulators for (eight; STDDEV_SAMP takes two) run in several passes over the same GROUP BY
    // columns: the groups and their order -- first occurrence, or key order -- do not depend on the aggregates, so pass k just fills its
    // own result columns (aggregate_hash.cpp:1016-1176 has one context per aggregate as well).  Each pass takes what fits.
    auto accumulators_of = [&](const hy_aggregate_spec& spec) -> uint32_t {
      return spec.function == HY_AGG_STDDEV_SAMP ? 2u : spec.function == HY_AGG_COUNT_DISTINCT ? 0u : 1u;
    };
    uint32_t needed = 0;
    for (uint32_t g = 0; g < n_aggregates; ++g) needed += accumulators_of(plain_aggregates[g]);
    if (needed <= MAX_AGGREGATES && n_aggregates <= MAX_AGGREGATES) return run_aggregate(plain_groupby.data(), n_groupby, plain_aggregates.data(), n_aggregates, out);
    uint32_t n_groups = 0;
    std::vector<hy_row_id> first_pass_rows;
    for (uint32_t begin = 0; begin < n_aggregates;) {
      uint32_t end = begin, taken = 0;
      while (end < n_aggregates && end - begin < MAX_AGGREGATES && taken + accumulators_of(plain_aggregates[end]) <= MAX_AGGREGATES) taken += accumulators_of(plain_aggregates[end++]);
      hy_aggregate_result pass = *out;
      pass.columns = out->columns + begin;
      std::vector<hy_row_id> rows;
      if (begin != 0) {   // (later passes: their representative rows only confirm that the groups came in the same order)
        rows.resize(out->group_capacity ? out->group_capacity : 1);
        pass.group_row_ids = out->group_row_ids ? rows.data() : nullptr;
      }
      const hy_status status = run_aggregate(plain_groupby.data(), n_groupby, plain_aggregates.data() + begin, end - begin, &pass);
      out->n_groups = pass.n_groups;
      if (status != HY_OK) return status;
      if (begin == 0) n_groups = pass.n_groups;
      else if (pass.n_groups != n_groups || (out->group_row_ids && n_groups && std::memcmp(rows.data(), out->group_row_ids, sizeof(hy_row_id) * n_groups) != 0))
        return fail(HY_ERR_DEVICE, "hy_aggregate_hash: two passes over the same GROUP BY columns disagree about the groups");
      begin = end;
    }
    return HY_OK;
  });
}

// debug only: which path the last hy_aggregate_hash of this process took -- 0 aggregate_rows, else the partition bits; not part of the public header
#ifdef HY_AGG_NEXT
#define HY_AGG_DEBUG(name, value) int HY_AGG_ENTRY(name)(void) { return g_last_aggregate_was_wide ? HY_AGG_NEXT(name)() : static_cast<int>(value); }
#else
#define HY_AGG_DEBUG(name, value) int HY_AGG_ENTRY(name)(void) { return static_cast<int>(value); }
#endif
HY_AGG_DEBUG(hy_debug_aggregate_path, g_agg_path)

// debug / tests only: 1 = the last aggregate of this process was ordered and written by the finish kernels (large results of plain functions)
HY_AGG_DEBUG(hy_debug_aggregate_finished_on_device, g_agg_finished_on_device)

// debug / tests only: 1 = the last hy_aggregate_hash of this process ran aggregate_small_domain (aggregate_small.hpp)
HY_AGG_DEBUG(hy_debug_aggregate_small_domain, g_agg_small)

// debug only (HY_AGG_TRACE): the per-slice phase stamps of the last aggregate_rows launch; not part of the public header
int HY_AGG_ENTRY(hy_debug_aggregate_trace)(uint64_t* out, uint32_t capacity_slices) {
  if (!g_agg_trace) return 0;
  const uint32_t n = g_agg_trace_slices < capacity_slices ? g_agg_trace_slices : capacity_slices;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(out, g_agg_trace, size_t{n} * 96, hipMemcpyDeviceToHost);
  return static_cast<int>(n);
}

}  // extern "C"
