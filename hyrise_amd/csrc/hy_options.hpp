// hy_options.hpp -- the library's process-wide options (include/hyrise_amd.h HY_OPT_*): which of several equivalent paths / launch
// shapes an operator takes.  An array of atomics filled with the defaults; hy_set_option stores, the operators load.  No getenv in a
// release build: the debug aids (traces, host timelines, kernels with parts switched off -- "results are wrong then") exist only
// under -DHY_DEBUG_SWITCHES (tools/build_variant.sh debug -DHY_DEBUG_SWITCHES), where HY_DEBUG_ENV reads the environment.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "../../include/hyrise_amd.h"

namespace hy {

extern std::atomic<int64_t> g_options[HY_OPT_COUNT];   // runtime.hip
inline int64_t option(uint32_t id) { return g_options[id].load(std::memory_order_relaxed); }

// What rounds 4-5 exposed as options and round 6 fixed (every setting gave the same results; the A/B timings that chose these values are in
// profiles/r03_scan_stores.txt, r04_join_variants.txt, r05_step_ab.txt and DESIGN.md).
constexpr int64_t FIXED_SCAN_WGS_PER_CU = 8;        // resident scan workgroups per CU (upper bound)
constexpr int64_t FIXED_SCAN_NT_STORES = 0;         // scan_slices writes its RowIDs with write-back stores
constexpr int64_t FIXED_PART_SLICES = 0;            // one scan part per chunk
constexpr int64_t FIXED_JOIN_IDENTITY = 1;          // a sorted build column of equally sized chunks is read in place (rank = row number)
constexpr int64_t FIXED_JOIN_FETCH_AHEAD = 1;       // probe segments are read through SliceViews (wide loads)
constexpr int64_t FIXED_JOIN_ORDERED_ATOMICS = 1;   // rank pairs with one returning LDS atomic each where the LDS serves lanes in order (probed once per process)
constexpr int64_t FIXED_JOIN_STORES = 2;            // pk_emit: write-back stores for the lines runs share, nontemporal ones in between
constexpr int64_t FIXED_JOIN_WGS_PER_CU = 0;        // persistent probe kernels: what the occupancy query says
constexpr int64_t FIXED_JOIN_EMIT_TILE_GROUP = 64;  // pk_emit: consecutive tiles per XCD (one front of 8 x 64 tiles moves through the probe side)
constexpr int64_t FIXED_JOIN_CLEAN_TABLES = 1;      // a hinted build's table and filter come zeroed: the join before cleared them
constexpr int64_t FIXED_SCAN_JOB_CACHE = 1;         // a data column remembers the per-chunk jobs of its last four literal predicates
constexpr int64_t FIXED_AGG_PARTITIONS = 1;         // many groups take the hash-partitioned path
constexpr int64_t FIXED_AGG_LDS_BUDGET = 32768;     // bytes of LDS a partition table may take
constexpr int64_t FIXED_AGG_SPLIT = 0;              // workgroups per partition: derived
constexpr int64_t FIXED_AGG_JOINT_HISTOGRAM = 1;    // two 1-byte measure columns are counted in one pair histogram
constexpr int64_t FIXED_FUSED_SHARED_PREFIX = 1;    // fused inputs that begin with an earlier input continue on its stack

}  // namespace hy

#ifdef HY_DEBUG_SWITCHES
#define HY_DEBUG_ENV(name) getenv(name)
#else
#define HY_DEBUG_ENV(name) (static_cast<const char*>(nullptr))
#endif
