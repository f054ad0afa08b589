// hy_scan_job.hpp -- the per-chunk normalised predicate of a TableScan (prepare_jobs in scan.hip writes it; scan_slices and
// the fused scan -> projection -> aggregate kernel of aggregate.hip evaluate it).
#pragma once
#include "hy_device.hpp"

namespace hy {

// ---- per-chunk normalised predicate ---------------------------------------------------------------------------------
enum : uint32_t { JOB_SCAN = 0, JOB_ALL = 1, JOB_NONE = 2,
                  JOB_RANGE = 3 /* a sorted chunk: the rows [range_begin, range_end) without [hole_begin, hole_end) match, nothing is read */ };
enum : uint32_t { KIND_U32 = 0, KIND_I64 = 1, KIND_F32 = 2, KIND_F64 = 3, KIND_NULLTEST = 4,
                  KIND_VISIBLE = 5 /* Validate: lo = snapshot commit id, span = our transaction id */,
                  KIND_VALUE_ID_SET = 6 /* LIKE family on dictionaries: lo = device address of the chunk's match bitmap */ };
enum : uint32_t { JF_INVERT = 1, JF_LOWER_INCL = 2, JF_UPPER_INCL = 4, JF_NEVER = 8 };

struct ScanJob {
  uint32_t mode;       // JOB_*
  uint32_t kind;       // KIND_*
  uint32_t flags;      // JF_*
  uint32_t null_vid;   // dictionary: value id that encodes NULL (aux_size); else 0xFFFFFFFF
  uint64_t lo;         // integer lower bound (bit pattern) | float/double lower bound bits
  uint64_t span;       // integer hi - lo                   | float/double upper bound bits
  // JOB_RANGE (sorted_segment_search.hpp): lo = range_begin | range_end << 32, span = hole_begin | hole_end << 32 (chunk offsets;
  // NotEquals leaves two ranges, :259-320) -- the struct stays 32 bytes: the streaming scan keeps two of them in scalar registers
};
__host__ __device__ inline uint32_t job_range_begin(const ScanJob& job) { return static_cast<uint32_t>(job.lo); }
__host__ __device__ inline uint32_t job_range_end(const ScanJob& job) { return static_cast<uint32_t>(job.lo >> 32); }
__host__ __device__ inline uint32_t job_hole_begin(const ScanJob& job) { return static_cast<uint32_t>(job.span); }
__host__ __device__ inline uint32_t job_hole_end(const ScanJob& job) { return static_cast<uint32_t>(job.span >> 32); }

struct PredicateArgs {
  uint32_t condition;
  uint32_t value_type;
  hy_value value;
  hy_value value2;
  const uint32_t* per_chunk_lower;
  const uint32_t* per_chunk_upper;
  const uint8_t* per_chunk_found;
  const uint64_t* match_words;          // LIKE family: per data chunk a bitmap over the dictionary's value ids
  const uint64_t* match_word_offsets;
  uint32_t column_is_nullable;
  uint32_t materialize_all;
  uint32_t no_ranges;                   // the caller evaluates the jobs itself and knows JOB_SCAN / JOB_ALL / JOB_NONE only (fused_rows)
};

// scan.hip: checks `predicate` for `column` (a data column) like hy_table_scan does, uploads its per-chunk arrays into
// `staging` (device memory of at least scan_jobs_staging_bytes()) and launches prepare_jobs on the current stream:
// jobs[c] = the normalised test of chunk c.
size_t scan_jobs_staging_bytes(const hy_column* column, const hy_predicate* predicate);
hy_status prepare_scan_jobs(const hy_column* column, const hy_predicate* predicate, ScanJob* jobs, void* staging);
// ... the same for Validate (hy_validate's jobs over a column of HY_ENC_MVCC segments): KIND_VISIBLE, or JOB_ALL for entirely visible chunks
hy_status prepare_visibility_scan_jobs(const hy_column* mvcc, uint32_t our_tid, uint32_t snapshot_commit_id, uint32_t can_use_chunk_shortcut, ScanJob* jobs, void* staging);

}  // namespace hy
