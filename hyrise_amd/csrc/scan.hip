// scan.hip -- TableScan on MI355X: ColumnVsValue / ColumnBetween / ColumnIsNull / ColumnVsColumn over Value,
// Dictionary and FrameOfReference segments (and reference segments through their pos lists).
//
// What it replaces (reference, CPU):
//   TableScan::_on_execute's per-chunk JobTask fan-out                     operators/table_scan.cpp:97-240
//   AbstractTableScanImpl::_scan_with_iterators (the hot loop)             table_scan/abstract_table_scan_impl.hpp:44-242
//   ColumnVsValue / ColumnBetween / ColumnIsNull / ColumnVsColumn impls    table_scan/column_*_table_scan_impl.cpp
//
// Device design (one launch for ALL chunks of a column; a chunk is ~128 KiB, far too small for a launch of its own):
//   prepare_jobs  one workgroup per data chunk: the dictionary bound searches (cooperative 64-ary) and the all/none early-outs of the
//                 reference (column_vs_value_table_scan_impl.cpp:211-272, column_between_table_scan_impl.cpp:112-170)
//                 collapse every predicate into ONE normalised per-chunk test:
//                     integers / value ids:  ((u)(x - lo) <= span) ^ invert        (type_comparison.hpp:120-132)
//                     float / double:        lower/upper compares with inclusive flags, ^ invert
//                     null test:             bitmap bit / value id == null id
//   scan_slices   persistent; one workgroup per PART (<= 8 slices of 8192 rows of ONE chunk -- a Hyrise chunk is one part).
//                 Per slice: 16-byte loads issued one slice ahead, a packed two-rows-per-instruction range test, one
//                 DPP prefix scan, compaction of the row numbers through LDS and line-aligned nontemporal 16-byte
//                 stores of RowID pairs into the chunk's own output region (region c starts at row_base[c], so no
//                 global prefix sum and no inter-workgroup traffic).  PosLists come out per chunk, ascending --
//                 bit-identical to what the CPU loop appends -- with the column read exactly once.
//   compact_regions  (host results only) packs the regions back to back.
// HBM-bound integer work: no MFMA anywhere.
#include "hy_device.hpp"
#include "hy_scan_job.hpp"

#include <cmath>
#include <cstring>
#include <limits>

namespace hy {

__device__ __forceinline__ bool is_between(uint32_t c) { return c >= HY_PRED_BETWEEN_INCLUSIVE && c <= HY_PRED_BETWEEN_EXCLUSIVE; }
__device__ __forceinline__ bool lower_inclusive(uint32_t c) { return c == HY_PRED_BETWEEN_INCLUSIVE || c == HY_PRED_BETWEEN_UPPER_EXCLUSIVE; }
__device__ __forceinline__ bool upper_inclusive(uint32_t c) { return c == HY_PRED_BETWEEN_INCLUSIVE || c == HY_PRED_BETWEEN_LOWER_EXCLUSIVE; }

// Wave-cooperative 64-ary search in a sorted dictionary: first index whose element is NOT (< value) [upper == false]
// or NOT (<= value) [upper == true]; d if there is none.  Two dependent loads for d <= 4096, three up to 262 144.
template <typename T>
__device__ uint32_t wave_bound(const T* dict, uint32_t d, T value, bool upper, uint32_t lane) {
  uint32_t lo = 0, hi = d;
  while (hi > lo) {
    const uint32_t span = hi - lo;
    const uint32_t step = (span + 63) / 64;
    const uint64_t idx = static_cast<uint64_t>(lo) + static_cast<uint64_t>(lane) * step;
    bool before = false;
    if (idx < hi) {
      const T e = dict[idx];
      before = upper ? !(value < e) : (e < value);
    }
    const uint32_t n = __popcll(__ballot(before));   // the probes that satisfy the predicate form a prefix
    if (n == 0) { hi = lo; break; }
    const uint64_t new_hi = static_cast<uint64_t>(lo) + static_cast<uint64_t>(n) * step;
    lo = lo + (n - 1) * step + 1;
    hi = new_hi < hi ? static_cast<uint32_t>(new_hi) : hi;
  }
  return lo;
}

template <typename T>
__device__ uint32_t wave_search(const DevSegment& s, T v, T v2, uint32_t which, uint32_t lane) {
  const T* dict = static_cast<const T*>(s.aux);
  const uint32_t r = wave_bound<T>(dict, s.aux_size, (which & 2) ? v2 : v, (which & 1) != 0, lane);
  return r == s.aux_size ? HY_INVALID_VALUE_ID : r;
}

template <typename T>
__device__ bool dict_equals(const DevSegment& s, uint32_t vid, T v) {
  return vid != HY_INVALID_VALUE_ID && static_cast<const T*>(s.aux)[vid] == v;
}

__device__ __forceinline__ void set_value_id_range(ScanJob& job, uint32_t lo, uint32_t hi_inclusive, bool invert) {
  job.kind = KIND_U32;
  job.lo = lo;
  job.span = hi_inclusive - lo;
  job.flags = invert ? JF_INVERT : 0;
}

// Signed integer predicate -> [lo, hi] in a wider type; returns false for an empty range.
template <typename Wide>
__device__ bool integer_range(uint32_t cond, Wide v, Wide v2, Wide tmin, Wide tmax, Wide* lo, Wide* hi, bool* invert) {
  *invert = false;
  switch (cond) {
    case HY_PRED_EQUALS: *lo = v; *hi = v; break;
    case HY_PRED_NOT_EQUALS: *lo = v; *hi = v; *invert = true; break;
    case HY_PRED_LESS_THAN: *lo = tmin; *hi = v - 1; break;
    case HY_PRED_LESS_THAN_EQUALS: *lo = tmin; *hi = v; break;
    case HY_PRED_GREATER_THAN: *lo = v + 1; *hi = tmax; break;
    case HY_PRED_GREATER_THAN_EQUALS: *lo = v; *hi = tmax; break;
    default:  // between: column_between_table_scan_impl.cpp:86-94 + type_comparison.hpp:120-132
      *lo = lower_inclusive(cond) ? v : v + 1;
      *hi = upper_inclusive(cond) ? v2 : v2 - 1;
      break;
  }
  return *lo <= *hi;
}

// ---- predicate -> ScanJob of one chunk --------------------------------------------------------------------------------------
// (Deriving the job at the head of each part inside scan_slices instead was measured: the four dependent loads cost the
// scan kernel what the separate launch costs the stream, ~4 us either way, so the separate kernel stays.)
//   job_search   wave w runs one of the (up to) four dictionary searches -- lower/upper bound of value and of value2 -- as a
//                cooperative 64-ary search, so a chunk costs two or three dependent loads instead of ~50
//   finish_job   the scalar rules of the reference's scan implementations on the four bounds
__device__ __forceinline__ bool job_searches(const DevSegment& seg, uint32_t cond, uint32_t wave) {
  // (an EMPTY dictionary -- every row NULL -- has no buffer either, but nothing to resolve: it is searched, and finds nothing)
  const bool searchable = seg.encoding == HY_ENC_DICTIONARY && (seg.aux || seg.aux_size == 0) && seg.data_type != HY_TYPE_STRING &&
                          cond != HY_PRED_IS_NULL && cond != HY_PRED_IS_NOT_NULL;
  return searchable && (wave < 2 || is_between(cond));
}

__device__ __forceinline__ uint32_t job_search(const DevSegment& seg, const PredicateArgs& p, uint32_t wave, uint32_t lane) {
  switch (seg.data_type) {
    case HY_TYPE_INT: return wave_search<int32_t>(seg, p.value.i32, p.value2.i32, wave, lane);
    case HY_TYPE_LONG: return wave_search<int64_t>(seg, p.value.i64, p.value2.i64, wave, lane);
    case HY_TYPE_FLOAT: return wave_search<float>(seg, p.value.f32, p.value2.f32, wave, lane);
    default: return wave_search<double>(seg, p.value.f64, p.value2.f64, wave, lane);
  }
}

__device__ __forceinline__ void finish_job(const DevSegment& s, uint32_t c, const PredicateArgs& p, const uint32_t* s_bound, ScanJob& job) {
  job.mode = JOB_SCAN;
  job.kind = KIND_U32;
  job.flags = 0;
  job.null_vid = 0xFFFFFFFFu;
  job.lo = 0;
  job.span = 0;
  const uint32_t cond = p.condition;

  if (cond >= HY_PRED_LIKE && cond <= HY_PRED_NOT_LIKE_INSENSITIVE) {   // column_like_table_scan_impl.cpp:69-121
    const uint64_t* bitmap = p.match_words + p.match_word_offsets[c];
    uint32_t matching = 0;
    for (uint32_t w = 0; w < (s.aux_size + 63) / 64; ++w) matching += __popcll(bitmap[w]);
    job.kind = KIND_VALUE_ID_SET;
    job.null_vid = s.aux_size;
    job.lo = reinterpret_cast<uint64_t>(bitmap);
    if (matching == 0) job.mode = JOB_NONE;   // "LIKE matches no rows"; "matches all rows" still tests every row for NULL
    return;
  }

  if (cond == HY_PRED_IS_NULL || cond == HY_PRED_IS_NOT_NULL) {
    const bool is_null = cond == HY_PRED_IS_NULL;
    if (s.encoding == HY_ENC_DICTIONARY) {  // column_is_null_table_scan_impl.cpp:167-195
      const bool all = is_null ? s.aux_size == 0 : s.aux_size == s.size;
      const bool none = is_null ? s.aux_size == s.size : s.aux_size == 0;
      if (all) job.mode = JOB_ALL;
      else if (none) job.mode = JOB_NONE;
      set_value_id_range(job, s.aux_size, s.aux_size, !is_null);
    } else {                                // :197-251 (nullable == has a null vector)
      if (!s.nulls) job.mode = is_null ? JOB_NONE : JOB_ALL;
      job.kind = KIND_NULLTEST;
      job.flags = is_null ? 0 : JF_INVERT;
    }
    return;
  }

  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t d = s.aux_size;
    job.null_vid = d;
    uint32_t lower = 0, upper = 0, lower2 = 0, upper2 = 0;
    bool found = false;
    if ((s.aux || s.aux_size == 0) && s.data_type != HY_TYPE_STRING) {
      lower = s_bound[0]; upper = s_bound[1]; lower2 = s_bound[2]; upper2 = s_bound[3];
      switch (s.data_type) {
        case HY_TYPE_INT: found = dict_equals<int32_t>(s, lower, p.value.i32); break;
        case HY_TYPE_LONG: found = dict_equals<int64_t>(s, lower, p.value.i64); break;
        case HY_TYPE_FLOAT: found = dict_equals<float>(s, lower, p.value.f32); break;
        default: found = dict_equals<double>(s, lower, p.value.f64); break;
      }
    } else {  // caller-resolved value ids (string dictionaries)
      lower = p.per_chunk_lower[c];
      upper = p.per_chunk_upper[c];
      found = p.per_chunk_found ? p.per_chunk_found[c] != 0 : false;
    }
    if (is_between(cond)) {  // column_between_table_scan_impl.cpp:112-193
      uint32_t lower_vid, upper_vid;
      if ((s.aux || s.aux_size == 0) && s.data_type != HY_TYPE_STRING) {
        lower_vid = lower_inclusive(cond) ? lower : upper;
        upper_vid = upper_inclusive(cond) ? upper2 : lower2;
      } else {
        lower_vid = lower;  // caller applied the inclusive/exclusive rule
        upper_vid = upper;
      }
      if (lower_vid == 0 && upper_vid == HY_INVALID_VALUE_ID) {
        if (p.column_is_nullable) {
          if (d == 0) job.flags = JF_NEVER; else set_value_id_range(job, 0, d - 1, false);
        } else {
          job.mode = JOB_ALL;
        }
      } else if (lower_vid == HY_INVALID_VALUE_ID || lower_vid >= upper_vid) {
        job.mode = JOB_NONE;
      } else {
        if (upper_vid == HY_INVALID_VALUE_ID) upper_vid = d;
        set_value_id_range(job, lower_vid, upper_vid - 1, false);
      }
    } else {                 // column_vs_value_table_scan_impl.cpp:89-272
      const bool use_upper = cond == HY_PRED_LESS_THAN_EQUALS || cond == HY_PRED_GREATER_THAN;
      const uint32_t search = use_upper ? upper : lower;
      bool all = false, none = false;
      switch (cond) {
        case HY_PRED_EQUALS: all = found && d == 1; none = !found; break;
        case HY_PRED_NOT_EQUALS: all = !found; none = found && d == 1; break;
        case HY_PRED_LESS_THAN:
        case HY_PRED_LESS_THAN_EQUALS: all = search == HY_INVALID_VALUE_ID; none = search == 0; break;
        default: all = search == 0; none = search == HY_INVALID_VALUE_ID; break;
      }
      if (all) {
        if (p.column_is_nullable) {  // still filter NULLs (:125-132)
          if (d == 0) job.flags = JF_NEVER; else set_value_id_range(job, 0, d - 1, false);
        } else {
          job.mode = JOB_ALL;
        }
      } else if (none) {
        job.mode = JOB_NONE;
      } else {
        // .hpp:57-81; NULL (== d) is outside every range below
        uint32_t range_lo = search, range_hi = search;
        bool invert = false;
        if (cond == HY_PRED_NOT_EQUALS) {
          invert = true;   // + explicit NULL check in the kernel
        } else if (cond == HY_PRED_LESS_THAN || cond == HY_PRED_LESS_THAN_EQUALS) {
          range_lo = 0;
          range_hi = search - 1;
        } else if (cond == HY_PRED_GREATER_THAN || cond == HY_PRED_GREATER_THAN_EQUALS) {
          range_hi = d - 1;
        }
        set_value_id_range(job, range_lo, range_hi, invert);
      }
    }
    return;
  }

  // Value / FrameOfReference segments: decode + typed compare with NULL check (_scan_generic_segment).
  switch (s.data_type) {
    case HY_TYPE_INT: {
      int64_t lo, hi;
      bool invert;
      if (!integer_range<int64_t>(cond, p.value.i32, p.value2.i32, INT32_MIN, INT32_MAX, &lo, &hi, &invert)) {
        job.flags = JF_NEVER;
      } else {
        job.kind = KIND_U32;
        job.lo = static_cast<uint32_t>(static_cast<int32_t>(lo));
        job.span = static_cast<uint32_t>(static_cast<uint64_t>(hi - lo));
        job.flags = invert ? JF_INVERT : 0;
      }
      break;
    }
    case HY_TYPE_LONG: {
      __int128 lo, hi;
      bool invert;
      if (!integer_range<__int128>(cond, p.value.i64, p.value2.i64, INT64_MIN, INT64_MAX, &lo, &hi, &invert)) {
        job.flags = JF_NEVER;
      } else {
        job.kind = KIND_I64;
        job.lo = static_cast<uint64_t>(static_cast<int64_t>(lo));
        job.span = static_cast<uint64_t>(hi - lo);
        job.flags = invert ? JF_INVERT : 0;
      }
      break;
    }
    case HY_TYPE_FLOAT:
    case HY_TYPE_DOUBLE: {
      const bool is_f32 = s.data_type == HY_TYPE_FLOAT;
      double a, b;
      uint32_t flags = 0;
      const double v = is_f32 ? static_cast<double>(p.value.f32) : p.value.f64;
      const double v2 = is_f32 ? static_cast<double>(p.value2.f32) : p.value2.f64;
      const double inf = INFINITY;
      switch (cond) {
        case HY_PRED_EQUALS: a = v; b = v; flags = JF_LOWER_INCL | JF_UPPER_INCL; break;
        case HY_PRED_NOT_EQUALS: a = v; b = v; flags = JF_LOWER_INCL | JF_UPPER_INCL | JF_INVERT; break;
        case HY_PRED_LESS_THAN: a = -inf; b = v; flags = JF_LOWER_INCL; break;
        case HY_PRED_LESS_THAN_EQUALS: a = -inf; b = v; flags = JF_LOWER_INCL | JF_UPPER_INCL; break;
        case HY_PRED_GREATER_THAN: a = v; b = inf; flags = JF_UPPER_INCL; break;
        case HY_PRED_GREATER_THAN_EQUALS: a = v; b = inf; flags = JF_LOWER_INCL | JF_UPPER_INCL; break;
        default:
          a = v; b = v2;
          flags = (lower_inclusive(cond) ? JF_LOWER_INCL : 0) | (upper_inclusive(cond) ? JF_UPPER_INCL : 0);
          break;
      }
      if (is_f32) {
        job.kind = KIND_F32;
        job.lo = __float_as_uint(static_cast<float>(a));
        job.span = __float_as_uint(static_cast<float>(b));
      } else {
        job.kind = KIND_F64;
        job.lo = static_cast<uint64_t>(__double_as_longlong(a));
        job.span = static_cast<uint64_t>(__double_as_longlong(b));
      }
      job.flags = flags;
      break;
    }
    default: job.mode = JOB_NONE; break;
  }
}

// ---- sorted chunks: SortedSegmentSearch (sorted_segment_search.hpp:20-384, used by column_vs_value_table_scan_impl.cpp:46-55,182-209 and
// column_between_table_scan_impl.cpp:60-110 when the chunk is flagged as sorted by the scanned column) -----------------------------------
// The reference narrows [begin, end) with lower_bound / upper_bound on the segment's values and writes the positions in between (two
// ranges for NotEquals).  Here: the chunk's normalised job classifies a row as below / inside / above the predicate's value range, a
// wave finds the ends of the NULL block and of the inside block with 64-ary searches (three dependent loads for 65 535 rows), and the
// job becomes JOB_RANGE: the scan kernel emits the positions without reading the segment.  The rows are the reference's: on a chunk
// that is sorted as flagged, "the rows between the bounds" and "the rows that satisfy the predicate" are the same set.
template <bool COMPRESSED>
__device__ __forceinline__ uint32_t load_element(const DevSegment& s, uint32_t i);   // (defined with the row evaluation below)
__device__ __forceinline__ uint32_t run_of_row(const DevSegment& s, uint32_t row);
__device__ __forceinline__ bool sorted_row_is_null(const DevSegment& s, uint32_t row) {
  if (s.encoding == HY_ENC_DICTIONARY) return load_element<true>(s, row) >= s.aux_size;
  if (s.encoding == HY_ENC_RUN_LENGTH) return s.nulls && reinterpret_cast<const uint8_t*>(s.nulls)[run_of_row(s, row)] != 0;
  return s.nulls && ((s.nulls[row >> 6] >> (row & 63)) & 1) != 0;
}

// -1: the row's value lies below the job's range, 0: inside, +1: above (the row is not NULL).
__device__ __forceinline__ int sorted_row_side(const DevSegment& s, const ScanJob& job, uint32_t row) {
  if (s.encoding == HY_ENC_DICTIONARY) {   // value ids: unsigned, ordered like the values
    const uint64_t x = load_element<true>(s, row), lo = static_cast<uint32_t>(job.lo), hi = lo + static_cast<uint32_t>(job.span);
    return x < lo ? -1 : x > hi ? 1 : 0;
  }
  const void* values = s.data;
  uint32_t index = row;
  if (s.encoding == HY_ENC_RUN_LENGTH) index = run_of_row(s, row);
  switch (job.kind) {
    case KIND_U32: {   // int32 values
      const int64_t v = s.encoding == HY_ENC_FRAME_OF_REFERENCE ? static_cast<int32_t>(load_element<true>(s, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]))
                                                                : static_cast<const int32_t*>(values)[index];
      const int64_t lo = static_cast<int32_t>(static_cast<uint32_t>(job.lo)), hi = lo + static_cast<uint32_t>(job.span);
      return v < lo ? -1 : v > hi ? 1 : 0;
    }
    case KIND_I64: {
      const int64_t v = static_cast<const int64_t*>(values)[index], lo = static_cast<int64_t>(job.lo);
      return v < lo ? -1 : static_cast<uint64_t>(v) - job.lo > job.span ? 1 : 0;
    }
    case KIND_F32: {
      const float x = static_cast<const float*>(values)[index], a = __uint_as_float(static_cast<uint32_t>(job.lo)), b = __uint_as_float(static_cast<uint32_t>(job.span));
      if (!((job.flags & JF_LOWER_INCL) ? x >= a : x > a)) return -1;
      return ((job.flags & JF_UPPER_INCL) ? x <= b : x < b) ? 0 : 1;
    }
    default: {
      const double x = static_cast<const double*>(values)[index], a = __longlong_as_double(static_cast<long long>(job.lo)), b = __longlong_as_double(static_cast<long long>(job.span));
      if (!((job.flags & JF_LOWER_INCL) ? x >= a : x > a)) return -1;
      return ((job.flags & JF_UPPER_INCL) ? x <= b : x < b) ? 0 : 1;
    }
  }
}

// The end of the prefix of [begin, end) whose rows satisfy `pred` (true for a prefix, false behind it), found by one wave: every round
// probes 64 rows and keeps the 1/64 of the range in which the prefix ends.
template <typename Pred>
__device__ __forceinline__ uint32_t wave_prefix_end(uint32_t begin, uint32_t end, uint32_t lane, Pred pred) {
  while (end - begin > 64) {
    const uint32_t step = (end - begin + 63) / 64;
    const uint32_t probe = begin + lane * step;
    const unsigned long long holds = __ballot(probe < end && pred(probe));
    const uint32_t first_false = ~holds ? static_cast<uint32_t>(__ffsll(static_cast<long long>(~holds)) - 1) : 64u;
    if (first_false == 0) return begin;
    const uint32_t known_false = begin + first_false * step;   // (beyond `end` if every probe held)
    begin += (first_false - 1) * step + 1;                     // the last probe that held, plus one
    if (first_false < 64 && known_false < end) end = known_false;
  }
  const unsigned long long holds = __ballot(begin + lane < end && pred(begin + lane));
  const uint32_t first_false = ~holds ? static_cast<uint32_t>(__ffsll(static_cast<long long>(~holds)) - 1) : 64u;
  return begin + first_false;
}

__device__ __forceinline__ bool job_takes_sorted_search(const DevSegment& seg, const PredicateArgs& p, const ScanJob& job) {
  const uint32_t cond = p.condition;
  return seg_sorted_by(seg) != HY_SORT_NONE && !p.no_ranges && seg.size != 0 && job.mode == JOB_SCAN && !(job.flags & JF_NEVER) && cond <= HY_PRED_BETWEEN_EXCLUSIVE &&
         job.kind != KIND_NULLTEST && job.kind != KIND_VALUE_ID_SET;
}

// One 256-thread workgroup per DATA chunk of the scanned column (for reference columns: of the referenced column).
__global__ __launch_bounds__(256) void prepare_jobs(const DevSegment* segments, uint32_t n_chunks, PredicateArgs p, ScanJob* jobs, uint32_t* overflow) {
  __shared__ uint32_t s_bound[4];
  __shared__ ScanJob s_job;
  __shared__ uint32_t s_range[2];
  const uint32_t c = blockIdx.x;
  if (c == 0 && threadIdx.x == 0) *overflow = 0;
  if (c >= n_chunks) return;
  const DevSegment seg = segments[c];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (job_searches(seg, p.condition, wave)) {
    const uint32_t r = job_search(seg, p, wave, lane);
    if (lane == 0) s_bound[wave] = r;
  }
  __syncthreads();
  if (seg_sorted_by(seg) == HY_SORT_NONE) {   // (uniform: the common case leaves here)
    if (threadIdx.x != 0) return;
    ScanJob job;
    finish_job(seg, c, p, s_bound, job);
    jobs[c] = job;
    return;
  }
  if (threadIdx.x == 0) {
    ScanJob job;
    finish_job(seg, c, p, s_bound, job);
    s_job = job;
  }
  __syncthreads();
  const ScanJob job = s_job;
  if (!job_takes_sorted_search(seg, p, job)) {
    if (threadIdx.x == 0) jobs[c] = job;
    return;
  }
  if (wave < 2) {   // wave 0: where the inside block begins, wave 1: where it ends; both find the NULL block first
    const uint32_t sorted_by = seg_sorted_by(seg);
    const bool ascending = sorted_by == HY_SORT_ASCENDING_NULLS_FIRST || sorted_by == HY_SORT_ASCENDING_NULLS_LAST;
    const bool nulls_last = sorted_by == HY_SORT_ASCENDING_NULLS_LAST || sorted_by == HY_SORT_DESCENDING_NULLS_LAST;
    uint32_t begin = 0, end = seg.size;
    if (seg.encoding == HY_ENC_DICTIONARY || seg.nulls) {
      if (nulls_last) end = wave_prefix_end(0, seg.size, lane, [&](uint32_t row) { return !sorted_row_is_null(seg, row); });
      else begin = wave_prefix_end(0, seg.size, lane, [&](uint32_t row) { return sorted_row_is_null(seg, row); });
    }
    // ascending: below | inside | above; descending: above | inside | below
    const int before = ascending ? -1 : 1;
    uint32_t bound;
    if (wave == 0) bound = wave_prefix_end(begin, end, lane, [&](uint32_t row) { return sorted_row_side(seg, job, row) == before; });
    else bound = wave_prefix_end(begin, end, lane, [&](uint32_t row) { return sorted_row_side(seg, job, row) != -before; });
    if (lane == 0) s_range[wave] = bound;
    if (wave == 0 && lane == 0) { s_bound[0] = begin; s_bound[1] = end; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  ScanJob ranged = job;
  ranged.mode = JOB_RANGE;
  const uint32_t inside_begin = s_range[0], inside_end = s_range[1] < s_range[0] ? s_range[0] : s_range[1];
  if (job.flags & JF_INVERT) {   // NotEquals: the non-NULL rows without the inside block
    ranged.lo = static_cast<uint64_t>(s_bound[0]) | static_cast<uint64_t>(s_bound[1]) << 32;
    ranged.span = static_cast<uint64_t>(inside_begin) | static_cast<uint64_t>(inside_end) << 32;
  } else {
    ranged.lo = static_cast<uint64_t>(inside_begin) | static_cast<uint64_t>(inside_end) << 32;
    ranged.span = 0;
  }
  jobs[c] = ranged;
}

// Validate: one job per chunk of the MVCC column.  Entirely visible chunks (validate.cpp:57-68) need no row test.
__global__ void prepare_visibility_jobs(const DevSegment* segments, uint32_t n_chunks, uint32_t our_tid, uint32_t snapshot, uint32_t can_use_chunk_shortcut,
                                        ScanJob* jobs, uint32_t* overflow) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) *overflow = 0;
  if (c >= n_chunks) return;
  const DevSegment s = segments[c];
  ScanJob job;
  job.kind = KIND_VISIBLE;
  job.flags = 0;
  job.null_vid = 0xFFFFFFFFu;
  job.lo = snapshot;
  job.span = our_tid;
  const bool is_mutable = (s.ref_chunk_id >> 31) != 0;
  const uint32_t invalid_rows = s.ref_chunk_id & 0x7FFFFFFFu, max_begin_cid = s.aux_size;
  job.mode = (can_use_chunk_shortcut && !is_mutable && snapshot >= max_begin_cid && invalid_rows == 0) ? JOB_ALL : JOB_SCAN;
  jobs[c] = job;
}

// ---- row evaluation ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t load_compressed(const void* data, uint32_t width, uint32_t i) {
  if (width == 1) return static_cast<const uint8_t*>(data)[i];
  if (width == 2) return static_cast<const uint16_t*>(data)[i];
  return static_cast<const uint32_t*>(data)[i];
}

// Element `i` of a segment's attribute / offset vector: FixedWidthInteger, or (width 0) a BitPackingVector -- compact::vector<uint32_t, 0,
// uint64_t>: bits [i * b, (i + 1) * b) of a little-endian stream of 64-bit words, unpacked in registers (bitpacking_decompressor.hpp:35-37).
__device__ __noinline__ uint32_t load_packed_element(const uint64_t* words, uint32_t bits, uint32_t i) {   // (not inlined: the rare layout must not cost the common ones registers)
  const uint64_t at = uint64_t{i} * bits;
  const uint32_t shift = static_cast<uint32_t>(at & 63);
  uint64_t value = words[at >> 6] >> shift;
  if (shift + bits > 64) value |= words[(at >> 6) + 1] << (64 - shift);
  return static_cast<uint32_t>(value & ((1ull << bits) - 1));
}
// COMPRESSED = false: the caller knows the column holds no bit-packed vector and no run-length segment (the instantiations of the
// scan kernel that every ordinary column takes carry none of that code).
template <bool COMPRESSED = true>
__device__ __forceinline__ uint32_t load_element(const DevSegment& s, uint32_t i) {
  if (!COMPRESSED || !seg_is_packed(s)) return load_compressed(s.data, s.width, i);
  return load_packed_element(static_cast<const uint64_t*>(s.data), seg_bits(s), i);
}

// A row against a JOB_RANGE job (a sorted chunk's matching rows, found by prepare_jobs).
__device__ __forceinline__ bool row_in_job_range(const ScanJob& job, uint32_t row) {
  return row - job_range_begin(job) < job_range_end(job) - job_range_begin(job) && !(row - job_hole_begin(job) < job_hole_end(job) - job_hole_begin(job));
}
// The rows [row0, row0 + 8) against [begin, end): one bit per row.
__device__ __forceinline__ uint32_t rows_in_range8(uint32_t row0, uint32_t begin, uint32_t end) {
  const uint32_t first = begin > row0 ? (begin - row0 < 8 ? begin - row0 : 8u) : 0u, last = end > row0 ? (end - row0 < 8 ? end - row0 : 8u) : 0u;
  return last > first ? ((1u << last) - 1u) & ~((1u << first) - 1u) : 0u;
}
// The 32 rows a lane owns in a slice of a JOB_RANGE chunk (the streaming instantiations' layout: four groups of eight rows).
__device__ __forceinline__ uint32_t job_range_mask(uint64_t range, uint64_t hole, uint32_t first_row) {
  uint32_t mask = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t row0 = first_row + k * 512;
    mask |= (rows_in_range8(row0, static_cast<uint32_t>(range), static_cast<uint32_t>(range >> 32)) &
             ~rows_in_range8(row0, static_cast<uint32_t>(hole), static_cast<uint32_t>(hole >> 32))) << (8 * k);
  }
  return mask;
}

__device__ __forceinline__ bool float_in_range(float x, const ScanJob& job) {
  const float a = __uint_as_float(static_cast<uint32_t>(job.lo)), b = __uint_as_float(static_cast<uint32_t>(job.span));
  const bool lower_ok = (job.flags & JF_LOWER_INCL) ? x >= a : x > a;
  const bool upper_ok = (job.flags & JF_UPPER_INCL) ? x <= b : x < b;
  return lower_ok && upper_ok;
}
__device__ __forceinline__ bool double_in_range(double x, const ScanJob& job) {
  const double a = __longlong_as_double(static_cast<long long>(job.lo)), b = __longlong_as_double(static_cast<long long>(job.span));
  const bool lower_ok = (job.flags & JF_LOWER_INCL) ? x >= a : x > a;
  const bool upper_ok = (job.flags & JF_UPPER_INCL) ? x <= b : x < b;
  return lower_ok && upper_ok;
}

// values[index] of an unencoded vector (a ValueSegment's values, a RunLengthSegment's run values) against the job's range.
__device__ __forceinline__ bool value_in_job(const void* values, uint32_t index, const ScanJob& job) {
  switch (job.kind) {
    case KIND_U32: return (static_cast<const uint32_t*>(values)[index] - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
    case KIND_I64: return (static_cast<const uint64_t*>(values)[index] - job.lo) <= job.span;
    case KIND_F32: return float_in_range(static_cast<const float*>(values)[index], job);
    default: return double_in_range(static_cast<const double*>(values)[index], job);
  }
}

// RunLengthSegment: the run of `row` = the first run whose inclusive end position is >= row (run_length_segment_iterable.hpp:100-160).
__device__ __noinline__ uint32_t run_of_position(const uint32_t* ends, uint32_t n_runs, uint32_t row) {
  uint32_t low = 0, high = n_runs - 1;
  while (low < high) {
    const uint32_t middle = (low + high) / 2;
    if (ends[middle] >= row) high = middle; else low = middle + 1;
  }
  return low;
}
__device__ __forceinline__ uint32_t run_of_row(const DevSegment& s, uint32_t row) { return run_of_position(static_cast<const uint32_t*>(s.aux), s.aux_size, row); }
__device__ __forceinline__ bool eval_run(const DevSegment& s, const ScanJob& job, uint32_t run) {
  const bool is_null = s.nulls && reinterpret_cast<const uint8_t*>(s.nulls)[run] != 0;
  const bool invert = job.flags & JF_INVERT;
  if (job.kind == KIND_NULLTEST) return is_null != invert;
  return !is_null && value_in_job(s.data, run, job) != invert;
}

// Scalar evaluation of one row of a DATA segment: tails, unaligned buffers and pos-list gathers.
template <bool COMPRESSED>
__device__ bool eval_row(const DevSegment& s, const ScanJob& job, uint32_t row) {
  if (job.mode == JOB_ALL) return true;
  if (job.mode == JOB_NONE || (job.flags & JF_NEVER)) return false;
  if (job.mode == JOB_RANGE) return row_in_job_range(job, row);
  if (COMPRESSED && s.encoding == HY_ENC_RUN_LENGTH) return eval_run(s, job, run_of_row(s, row));
  if (s.encoding == HY_ENC_MVCC) {   // Validate::is_row_visible (validate.cpp:47-55)
    const uint32_t snapshot = static_cast<uint32_t>(job.lo), our_tid = static_cast<uint32_t>(job.span);
    const uint32_t tid = static_cast<const uint32_t*>(s.data)[row], begin = static_cast<const uint32_t*>(s.aux)[row];
    const uint32_t end = reinterpret_cast<const uint32_t*>(s.nulls)[row];
    return snapshot < end && ((snapshot >= begin) != (tid == our_tid));
  }
  const bool invert = job.flags & JF_INVERT;
  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = load_element<COMPRESSED>(s, row);
    if (job.kind == KIND_VALUE_ID_SET) {
      return vid < job.null_vid && ((reinterpret_cast<const uint64_t*>(job.lo)[vid >> 6] >> (vid & 63)) & 1) != 0;
    }
    const bool in = (vid - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
    return (in != invert) && vid != job.null_vid;   // null_vid is 0xFFFFFFFF for the NULL test itself
  }
  const bool is_null = s.nulls ? ((s.nulls[row >> 6] >> (row & 63)) & 1) != 0 : false;
  if (job.kind == KIND_NULLTEST) return is_null != invert;
  if (is_null) return false;
  bool in;
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    const uint32_t x = load_element<COMPRESSED>(s, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]);
    in = (x - static_cast<uint32_t>(job.lo)) <= static_cast<uint32_t>(job.span);
  } else {
    in = value_in_job(s.data, row, job);
  }
  return in != invert;
}

// 8 consecutive rows starting at row0 (multiple of 8), all valid, buffers 16-byte aligned: vector loads.
template <int WIDTH>
__device__ __forceinline__ void load8(const void* data, uint32_t row0, uint32_t (&x)[8]) {
  if constexpr (WIDTH == 1) {
    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const uint8_t*>(data) + row0);
    x[0] = v.x & 0xFF; x[1] = (v.x >> 8) & 0xFF; x[2] = (v.x >> 16) & 0xFF; x[3] = v.x >> 24;
    x[4] = v.y & 0xFF; x[5] = (v.y >> 8) & 0xFF; x[6] = (v.y >> 16) & 0xFF; x[7] = v.y >> 24;
  } else if constexpr (WIDTH == 2) {
    const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(data) + row0);
    x[0] = v.x & 0xFFFF; x[1] = v.x >> 16; x[2] = v.y & 0xFFFF; x[3] = v.y >> 16;
    x[4] = v.z & 0xFFFF; x[5] = v.z >> 16; x[6] = v.w & 0xFFFF; x[7] = v.w >> 16;
  } else {
    const uint4 v0 = *reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(data) + row0);
    const uint4 v1 = *reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(data) + row0 + 4);
    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
  }
}

template <int WIDTH>
__device__ __forceinline__ uint32_t eval8_u32(const DevSegment& s, const ScanJob& job, uint32_t row0, uint32_t bias) {
  uint32_t x[8];
  load8<WIDTH>(s.data, row0, x);
  const uint32_t lo = static_cast<uint32_t>(job.lo) - bias, span = static_cast<uint32_t>(job.span);
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) bits |= ((x[j] - lo) <= span ? 1u : 0u) << j;
  return bits;
}

// RunLengthSegment: one search of the end positions, then the runs are walked (run_length_segment_iterable.hpp:100-160).
__device__ __forceinline__ uint32_t eval8_run_length(const DevSegment& s, const ScanJob& job, uint32_t row0, uint32_t valid) {
  const uint32_t* ends = static_cast<const uint32_t*>(s.aux);
  uint32_t run = run_of_row(s, row0), bits = 0;
  bool matches = eval_run(s, job, run);
  for (uint32_t j = 0; j < valid; ++j) {
    if (row0 + j > ends[run]) { ++run; matches = eval_run(s, job, run); }
    bits |= (matches ? 1u : 0u) << j;
  }
  return bits;
}

// Match bits of 8 rows [row0, row0+8) of a data segment; rows >= size are masked off by the caller.
template <bool COMPRESSED>
__device__ __forceinline__ uint32_t eval8(const DevSegment& s, const ScanJob& job, uint32_t row0, uint32_t valid) {
  if (job.mode == JOB_RANGE) {
    return rows_in_range8(row0, job_range_begin(job), job_range_end(job)) & ~rows_in_range8(row0, job_hole_begin(job), job_hole_end(job)) & ((1u << valid) - 1u);
  }
  if (COMPRESSED && s.encoding == HY_ENC_RUN_LENGTH) return eval8_run_length(s, job, row0, valid);
  if (valid < 8 || (s.flags & SEG_UNALIGNED) || (COMPRESSED && seg_is_packed(s))) {
    uint32_t bits = 0;
    for (uint32_t j = 0; j < valid; ++j) bits |= (eval_row<COMPRESSED>(s, job, row0 + j) ? 1u : 0u) << j;
    return bits;
  }
  if (s.encoding == HY_ENC_MVCC) {   // eight rows: 2 x 16 B of each of the three arrays
    const uint32_t snapshot = static_cast<uint32_t>(job.lo), our_tid = static_cast<uint32_t>(job.span);
    uint32_t tid[8], begin[8], end[8];
    load8<4>(s.data, row0, tid);
    load8<4>(s.aux, row0, begin);
    load8<4>(s.nulls, row0, end);
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) bits |= ((snapshot < end[j] && ((snapshot >= begin[j]) != (tid[j] == our_tid))) ? 1u : 0u) << j;
    return bits;
  }
  const uint32_t inv = (job.flags & JF_INVERT) ? 0xFFu : 0u;
  if (s.encoding == HY_ENC_DICTIONARY && job.kind == KIND_VALUE_ID_SET) {
    uint32_t x[8], bits = 0;
    if (s.width == 2) load8<2>(s.data, row0, x);
    else if (s.width == 1) load8<1>(s.data, row0, x);
    else load8<4>(s.data, row0, x);
    const uint64_t* bitmap = reinterpret_cast<const uint64_t*>(job.lo);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t vid = x[j] < job.null_vid ? x[j] : 0;   // the NULL id has no bit
      bits |= ((x[j] < job.null_vid && ((bitmap[vid >> 6] >> (vid & 63)) & 1)) ? 1u : 0u) << j;
    }
    return bits;
  }
  if (s.encoding == HY_ENC_DICTIONARY) {
    uint32_t bits, nullbits = 0;
    uint32_t x[8];
    const uint32_t lo = static_cast<uint32_t>(job.lo), span = static_cast<uint32_t>(job.span);
    if (s.width == 2) load8<2>(s.data, row0, x);
    else if (s.width == 1) load8<1>(s.data, row0, x);
    else load8<4>(s.data, row0, x);
    bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) bits |= ((x[j] - lo) <= span ? 1u : 0u) << j;
    if (inv) {  // != : NULL never matches (column_vs_value_table_scan_impl.cpp:171-177); 0xFFFFFFFF for IS NOT NULL
#pragma unroll
      for (int j = 0; j < 8; ++j) nullbits |= (x[j] == job.null_vid ? 1u : 0u) << j;
    }
    return (bits ^ inv) & ~nullbits;
  }
  const uint32_t nullbits = s.nulls ? reinterpret_cast<const uint8_t*>(s.nulls)[row0 >> 3] : 0u;
  if (job.kind == KIND_NULLTEST) return (nullbits ^ inv) & 0xFFu;
  uint32_t bits = 0;
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    // value = offset + block minimum; test (offset + min - lo) <= span, i.e. compare offsets against (lo - min)
    const uint32_t bias = static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row0 / HY_FOR_BLOCK_SIZE]);
    if (s.width == 2) bits = eval8_u32<2>(s, job, row0, bias);
    else if (s.width == 1) bits = eval8_u32<1>(s, job, row0, bias);
    else bits = eval8_u32<4>(s, job, row0, bias);
  } else if (job.kind == KIND_U32) {
    bits = eval8_u32<4>(s, job, row0, 0);
  } else if (job.kind == KIND_I64) {
    const uint4* p = reinterpret_cast<const uint4*>(static_cast<const uint64_t*>(s.data) + row0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = p[q];
      const uint64_t a = (static_cast<uint64_t>(v.y) << 32) | v.x, b = (static_cast<uint64_t>(v.w) << 32) | v.z;
      bits |= ((a - job.lo) <= job.span ? 1u : 0u) << (2 * q);
      bits |= ((b - job.lo) <= job.span ? 1u : 0u) << (2 * q + 1);
    }
  } else if (job.kind == KIND_F32) {
    const uint4* p = reinterpret_cast<const uint4*>(static_cast<const float*>(s.data) + row0);
    const uint4 v0 = p[0], v1 = p[1];
    const uint32_t raw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) bits |= (float_in_range(__uint_as_float(raw[j]), job) ? 1u : 0u) << j;
  } else {
    const uint4* p = reinterpret_cast<const uint4*>(static_cast<const double*>(s.data) + row0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = p[q];
      const double a = __longlong_as_double((static_cast<long long>(v.y) << 32) | v.x);
      const double b = __longlong_as_double((static_cast<long long>(v.w) << 32) | v.z);
      bits |= (double_in_range(a, job) ? 1u : 0u) << (2 * q);
      bits |= (double_in_range(b, job) ? 1u : 0u) << (2 * q + 1);
    }
  }
  return (bits ^ inv) & ~nullbits & 0xFFu;
}

// Decoded numeric value of one row of a DATA segment (ColumnVsColumn), as the widest carrier of its class.
struct Cell {
  bool is_null;
  int64_t i;
  double f;
};
__device__ Cell load_cell(const DevSegment& s, uint32_t row) {
  Cell c{false, 0, 0.0};
  if (s.encoding == HY_ENC_DICTIONARY) {
    const uint32_t vid = load_compressed(s.data, s.width, row);
    if (vid >= s.aux_size) { c.is_null = true; return c; }
    switch (s.data_type) {
      case HY_TYPE_INT: c.i = static_cast<const int32_t*>(s.aux)[vid]; break;
      case HY_TYPE_LONG: c.i = static_cast<const int64_t*>(s.aux)[vid]; break;
      case HY_TYPE_FLOAT: c.f = static_cast<const float*>(s.aux)[vid]; break;
      default: c.f = static_cast<const double*>(s.aux)[vid]; break;
    }
    return c;
  }
  if (s.nulls && ((s.nulls[row >> 6] >> (row & 63)) & 1)) { c.is_null = true; return c; }
  if (s.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    c.i = static_cast<int32_t>(load_compressed(s.data, s.width, row) + static_cast<uint32_t>(static_cast<const int32_t*>(s.aux)[row / HY_FOR_BLOCK_SIZE]));
    return c;
  }
  switch (s.data_type) {
    case HY_TYPE_INT: c.i = static_cast<const int32_t*>(s.data)[row]; break;
    case HY_TYPE_LONG: c.i = static_cast<const int64_t*>(s.data)[row]; break;
    case HY_TYPE_FLOAT: c.f = static_cast<const float*>(s.data)[row]; break;
    default: c.f = static_cast<const double*>(s.data)[row]; break;
  }
  return c;
}

template <typename T>
__device__ __forceinline__ bool compare(uint32_t cond, T a, T b) {
  switch (cond) {
    case HY_PRED_EQUALS: return a == b;
    case HY_PRED_NOT_EQUALS: return a != b;
    case HY_PRED_LESS_THAN: return a < b;
    case HY_PRED_LESS_THAN_EQUALS: return a <= b;
    case HY_PRED_GREATER_THAN: return a > b;
    default: return a >= b;
  }
}

// Row `row` of chunk `chunk` of a column that may consist of reference segments.
__device__ Cell column_cell(const DevSegment* segments, uint32_t chunk, uint32_t row, uint32_t* type) {
  const DevSegment& s = segments[chunk];
  *type = s.data_type;
  if (s.encoding != HY_ENC_REFERENCE) return load_cell(s, row);
  hy_row_id r;
  if (s.data) r = static_cast<const hy_row_id*>(s.data)[row];
  else { r.chunk_id = s.ref_chunk_id; r.chunk_offset = row; }
  if (r.chunk_offset == 0xFFFFFFFFu) return Cell{true, 0, 0.0};
  return load_cell(s.ref[r.chunk_id], r.chunk_offset);
}

// C++ usual arithmetic conversions of `left OP right` (column_vs_column_table_scan_impl.cpp:169-184).
__device__ bool compare_cells(uint32_t cond, const Cell& l, uint32_t lt, const Cell& r, uint32_t rt) {
  const bool lf = lt == HY_TYPE_FLOAT || lt == HY_TYPE_DOUBLE, rf = rt == HY_TYPE_FLOAT || rt == HY_TYPE_DOUBLE;
  if (!lf && !rf) return compare<int64_t>(cond, l.i, r.i);
  if (lt == HY_TYPE_DOUBLE || rt == HY_TYPE_DOUBLE) return compare<double>(cond, lf ? l.f : static_cast<double>(l.i), rf ? r.f : static_cast<double>(r.i));
  return compare<float>(cond, lf ? static_cast<float>(l.f) : static_cast<float>(l.i), rf ? static_cast<float>(r.f) : static_cast<float>(r.i));
}

// ---- the scan kernel ----------------------------------------------------------------------------------------------------

struct ScanArgs {
  const DevSegment* segments;      // scanned column
  const DevSegment* right;         // ColumnVsColumn: right column, else nullptr
  const Slice* slices;
  const ScanJob* jobs;             // per DATA chunk (of the referenced column for reference columns)
  uint32_t n_slices;
  uint32_t n_parts;
  uint32_t n_chunks;
  uint32_t condition;              // ColumnVsColumn
  uint32_t materialize_all;
  uint32_t is_null_scan;           // IS NULL on reference columns: NULL_ROW_IDs match
  uint32_t epoch;
  uint64_t* status;                // [n_parts] epoch-tagged part totals (only touched by multi-part chunks)
  hy_row_id* matches;              // chunk regions
  uint64_t capacity;
  uint64_t* offsets;               // [n_chunks + 1] region starts
  uint32_t* counts;                // [n_chunks] or nullptr
  uint8_t* chunk_state;            // [n_chunks] or nullptr
  uint32_t* overflow;              // set to 1 if capacity was exceeded (a persistent, normally-zero word of the scratch)
  uint64_t* trace;                 // debug: 4 wall-clock stamps per workgroup (HY_SCAN_TRACE), else nullptr
  uint32_t plain_stores;           // write-back stores for the RowIDs (the default; HY_SCAN_NT_STORES=1: nontemporal ones, for A/B runs with tools/scan_ab.py)
};

__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(v, d, 64);
    if (lane >= static_cast<uint32_t>(d)) v += t;
  }
  return v;
}

// Match bits of 8 already-loaded 32-bit lanes.
__device__ __forceinline__ uint32_t range_bits8(const uint32_t (&x)[8], uint32_t lo, uint32_t span) {
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) bits |= ((x[j] - lo) <= span ? 1u : 0u) << j;
  return bits;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int W>
struct RawGroup;   // 8 rows of W-byte elements as loaded from HBM
template <> struct RawGroup<1> { u32x2 v; };
template <> struct RawGroup<2> { u32x4 v; };
template <> struct RawGroup<4> { u32x4 v0, v1; };
// W = 16: a BitPackingVector of at most 16 bits per element -- the group's b bytes start at byte (row0 / 8) * b of the word stream: the 20
// bytes from the 4-byte boundary below it hold them all
template <> struct RawGroup<16> { u32x4 v; uint32_t tail; uint32_t shift; };

// Segment buffers are always global memory: say so, or the compiler emits flat loads (which also tick lgkmcnt).
#define HY_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const HY_GLOBAL T* as_global(const void* p) {
  return (const HY_GLOBAL T*)(p);
}

template <int W>
__device__ __forceinline__ RawGroup<W> load_group(const void* data, uint32_t row0) {
  RawGroup<W> g;
  const HY_GLOBAL uint8_t* base = as_global<uint8_t>(data) + static_cast<size_t>(row0) * W;
  if constexpr (W == 1) {
    g.v = *(const HY_GLOBAL u32x2*)(base);
  } else if constexpr (W == 2) {
    g.v = *(const HY_GLOBAL u32x4*)(base);
  } else {
    g.v0 = *(const HY_GLOBAL u32x4*)(base);
    g.v1 = *(const HY_GLOBAL u32x4*)(base + 16);
  }
  return g;
}

__device__ __forceinline__ RawGroup<16> load_packed_group(const void* data, uint32_t row0, uint32_t bits) {
  RawGroup<16> g;
  const uint32_t byte_offset = (row0 >> 3) * bits;   // (eight rows are `bits` whole bytes)
  const HY_GLOBAL uint8_t* base = as_global<uint8_t>(data) + (byte_offset & ~3u);
  g.v = *(const HY_GLOBAL u32x4*)(base);
  g.tail = *(const HY_GLOBAL uint32_t*)(base + 16);
  g.shift = (byte_offset & 3u) * 8;
  return g;
}
// Eight elements of `bits` bits each (1 .. 16, the same for every lane): the window is shifted down to the group's first bit, then
// element j sits at bit j * bits -- a position every lane shares, so the word selects are scalar (bitpacking_decompressor.hpp:35-37).
__device__ __forceinline__ void unpack_packed_group(const RawGroup<16>& g, uint32_t bits, uint32_t (&x)[8]) {
  const uint32_t a0 = __builtin_amdgcn_alignbit(g.v.y, g.v.x, g.shift), a1 = __builtin_amdgcn_alignbit(g.v.z, g.v.y, g.shift);
  const uint32_t a2 = __builtin_amdgcn_alignbit(g.v.w, g.v.z, g.shift), a3 = __builtin_amdgcn_alignbit(g.tail, g.v.w, g.shift);
  const uint32_t mask = (1u << bits) - 1u;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t position = j * bits, word = position >> 5, shift = position & 31;
    const uint32_t low = word == 0 ? a0 : word == 1 ? a1 : word == 2 ? a2 : a3;
    const uint32_t high = word == 0 ? a1 : word == 1 ? a2 : word == 2 ? a3 : 0u;
    x[j] = __builtin_amdgcn_alignbit(high, low, shift) & mask;
  }
}

template <int W>
__device__ __forceinline__ void unpack_group(const RawGroup<W>& g, uint32_t (&x)[8]) {
  if constexpr (W == 1) {
    x[0] = g.v.x & 0xFF; x[1] = (g.v.x >> 8) & 0xFF; x[2] = (g.v.x >> 16) & 0xFF; x[3] = g.v.x >> 24;
    x[4] = g.v.y & 0xFF; x[5] = (g.v.y >> 8) & 0xFF; x[6] = (g.v.y >> 16) & 0xFF; x[7] = g.v.y >> 24;
  } else if constexpr (W == 2) {
    x[0] = g.v.x & 0xFFFF; x[1] = g.v.x >> 16; x[2] = g.v.y & 0xFFFF; x[3] = g.v.y >> 16;
    x[4] = g.v.z & 0xFFFF; x[5] = g.v.z >> 16; x[6] = g.v.w & 0xFFFF; x[7] = g.v.w >> 16;
  } else {
    x[0] = g.v0.x; x[1] = g.v0.y; x[2] = g.v0.z; x[3] = g.v0.w; x[4] = g.v1.x; x[5] = g.v1.y; x[6] = g.v1.z; x[7] = g.v1.w;
  }
}

// ---- ColumnVsColumn over 4-byte columns (TPC-H Q4 / Q12: l_commitdate < l_receiptdate) ------------------------------------------
// A segment whose rows decode to one 4-byte word each -- unencoded int32 / float, a dictionary of such values, or
// FrameOfReference -- and whose vector can be read in aligned groups of eight rows.
__device__ __forceinline__ bool four_byte_rows(const DevSegment& s) {
  if ((s.data_type != HY_TYPE_INT && s.data_type != HY_TYPE_FLOAT) || (s.flags & SEG_UNALIGNED)) return false;
  if (s.encoding == HY_ENC_UNENCODED) return true;
  if (s.encoding == HY_ENC_DICTIONARY || s.encoding == HY_ENC_FRAME_OF_REFERENCE) return s.width == 1 || s.width == 2 || s.width == 4;
  return false;
}

// Rows row0 .. row0 + 7 of such a segment (row0 a multiple of eight, at least one of the rows inside the segment): the eight
// stored elements come with ONE aligned load; dictionary value ids are then looked up (the dictionaries of a chunk are a few
// KiB: cache hits).  Bit j of *nulls: row j is NULL.
__device__ __forceinline__ void fetch_four_byte_rows(const DevSegment& s, uint32_t row0, uint32_t (&word)[8], uint32_t* nulls) {
  uint32_t x[8];
  if (s.encoding == HY_ENC_UNENCODED || s.width == 4) unpack_group<4>(load_group<4>(s.data, row0), x);
  else if (s.width == 2) unpack_group<2>(load_group<2>(s.data, row0), x);
  else unpack_group<1>(load_group<1>(s.data, row0), x);
  *nulls = 0;
  if (s.encoding == HY_ENC_DICTIONARY) {
    const HY_GLOBAL uint32_t* dictionary = as_global<uint32_t>(s.aux);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool is_null = x[j] >= s.aux_size;
      if (is_null) *nulls |= 1u << j;
      word[j] = dictionary[is_null ? 0u : x[j]];   // (a NULL-only chunk has an empty dictionary: the word is never used, the load stays inside the allocation's page)
    }
    if (s.aux_size == 0) *nulls = 0xFF;
    return;
  }
  if (s.nulls) *nulls = as_global<uint8_t>(s.nulls)[row0 >> 3];
  const uint32_t bias = s.encoding == HY_ENC_FRAME_OF_REFERENCE ? as_global<uint32_t>(s.aux)[row0 / HY_FOR_BLOCK_SIZE] : 0u;   // (eight aligned rows never straddle a 2048-row block)
#pragma unroll
  for (int j = 0; j < 8; ++j) word[j] = x[j] + bias;
}

// scan_slices (below): persistent grid, one workgroup per PART (<= 8 slices of one chunk; a Hyrise chunk is one part), ONE fused
// pass per 8192-row slice:
//   evaluate  32-bit match mask per lane (rows wave*2048 + k*512 + lane*8 + j).  Streaming instantiations (W = 1 / 2 / 4: every
//             segment of the column is a W-byte attribute vector / FoR offset vector / int32 value vector): the 16-byte loads of
//             the next slice -- also across parts -- are issued before this slice is evaluated.  W = 0 (evaluate_slice): every
//             other shape -- ColumnVsColumn, reference segments, 8-byte values, LIKE bitmaps, MVCC visibility.
//   count     masks transposed inside the wave so that a lane holds 32 consecutive rows, one DPP prefix scan, the four wave
//             totals meet in LDS (the only workgroup barrier per slice)
//   emit      each wave compacts its row numbers through private LDS and writes RowID pairs with line-aligned nontemporal
//             16-byte stores at the running offset of the chunk's region.
// The column is read from HBM exactly once and every RowID is written exactly once.  Output order is (chunk, row) ascending:
// bit-identical to the CPU loop's appends.  Chunks of more than one part chain their parts through one epoch-tagged status
// word per part.
template <bool COMPRESSED>
__device__ __forceinline__ uint32_t evaluate_slice(const ScanArgs& a, const Slice& slice, const DevSegment& seg, uint32_t wave, uint32_t lane) {
  uint32_t mask = 0;       // bit (8k + j) <-> row  wave*2048 + k*512 + lane*8 + j  of the slice
  const DevSegment right_seg = a.right ? a.right[slice.chunk] : DevSegment{};
  if (a.right && seg.data_type == right_seg.data_type && four_byte_rows(seg) && four_byte_rows(right_seg) && (seg.aux_size || seg.encoding != HY_ENC_DICTIONARY) &&
      (right_seg.aux_size || right_seg.encoding != HY_ENC_DICTIONARY)) {
    // ColumnVsColumn, both sides 4-byte rows of one type: eight rows per lane and step, each side one wide load (+ the
    // dictionary lookups); int32 against int32 and float against float compare like the reference's typed comparator
    // (column_vs_column_table_scan_impl.cpp:169-184)
    const bool is_float = seg.data_type == HY_TYPE_FLOAT;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
      if (r0 >= slice.row_count) continue;
      const uint32_t valid = (slice.row_count - r0 < 8) ? slice.row_count - r0 : 8;
      uint32_t left[8], right[8], left_nulls, right_nulls;
      fetch_four_byte_rows(seg, slice.row_begin + r0, left, &left_nulls);
      fetch_four_byte_rows(right_seg, slice.row_begin + r0, right, &right_nulls);
      uint32_t bits = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool m = is_float ? compare<float>(a.condition, __uint_as_float(left[j]), __uint_as_float(right[j]))
                                : compare<int32_t>(a.condition, static_cast<int32_t>(left[j]), static_cast<int32_t>(right[j]));
        bits |= (m ? 1u : 0u) << j;
      }
      mask |= (bits & ~(left_nulls | right_nulls) & ((1u << valid) - 1u)) << (8 * k);
    }
  } else if (a.right) {
    // ColumnVsColumn: both sides decoded per row (no SIMD path in the reference either, abstract_table_scan_impl.hpp:61-66)
#pragma unroll 1
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
      for (uint32_t j = 0; j < 8 && r0 + j < slice.row_count; ++j) {
        const uint32_t row = slice.row_begin + r0 + j;
        uint32_t lt, rt;
        const Cell l = column_cell(a.segments, slice.chunk, row, &lt);
        const Cell r = column_cell(a.right, slice.chunk, row, &rt);
        if (!l.is_null && !r.is_null && compare_cells(a.condition, l, lt, r, rt)) mask |= 1u << (8 * k + j);
      }
    }
  } else if (seg.encoding == HY_ENC_REFERENCE) {
    uint32_t mode = JOB_SCAN;
    if (seg.ref_chunk_id != 0xFFFFFFFFu) mode = a.jobs[seg.ref_chunk_id].mode;
    if ((mode == JOB_SCAN || mode == JOB_RANGE || (mode == JOB_ALL && a.materialize_all)) && seg.ref_chunk_id != 0xFFFFFFFFu) {
      // A PosList that references ONE chunk (what a first TableScan, a Validate or a join's write_output_chunks guarantee,
      // abstract_dereferenced_column_table_scan_impl.cpp:38-46): the referenced segment and its job are the same for every row --
      // scalar registers instead of two descriptor loads per row -- and only the offsets of the RowIDs are read.
      const DevSegment base = seg.ref[seg.ref_chunk_id];
      const ScanJob job = a.jobs[seg.ref_chunk_id];
      const HY_GLOBAL uint32_t* words = as_global<uint32_t>(seg.data);
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
        uint32_t offset[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t row = slice.row_begin + (r0 + j < slice.row_count ? r0 + j : 0);
          offset[j] = words ? words[2 * size_t{row} + 1] : row;
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          if (r0 + j >= slice.row_count) continue;
          const bool m = offset[j] == 0xFFFFFFFFu ? a.is_null_scan != 0 : eval_row<false>(base, job, offset[j]);   // (a reference segment points at the decoded twin)
          if (m) mask |= 1u << (8 * k + j);
        }
      }
    } else if (mode == JOB_SCAN || mode == JOB_RANGE || (mode == JOB_ALL && a.materialize_all)) {
#pragma unroll 1
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
        for (uint32_t j = 0; j < 8 && r0 + j < slice.row_count; ++j) {
          const uint32_t row = slice.row_begin + r0 + j;
          hy_row_id r;
          if (seg.data) r = static_cast<const hy_row_id*>(seg.data)[row];
          else { r.chunk_id = seg.ref_chunk_id; r.chunk_offset = row; }
          bool m;
          if (r.chunk_offset == 0xFFFFFFFFu) m = a.is_null_scan != 0;   // NULL_ROW_ID: only IS NULL matches
          else m = eval_row<false>(seg.ref[r.chunk_id], a.jobs[r.chunk_id], r.chunk_offset);
          if (m) mask |= 1u << (8 * k + j);
        }
      }
    }
  } else {
    const ScanJob job = a.jobs[slice.chunk];
    bool walked = false;
    if (COMPRESSED && job.mode == JOB_SCAN && !(job.flags & JF_NEVER) && seg.encoding == HY_ENC_RUN_LENGTH && wave * 2048 < slice.row_count) {
      // RunLengthSegment, run by run (run_length_segment_iterable.hpp:100-160 walks the runs; so does the wave): the runs that overlap the
      // wave's 2048 rows are found with one search of the end positions for the whole wave (scalar loads), each is tested once, and a
      // lane marks the rows of its four groups that lie in a matching run -- no lane reads anything.  (Clustered dates, 17 000 rows per run:
      // one search per eight rows was 98 us for 60 M rows, of which the 206 MB of positions take 49.)  Segments with short runs -- more
      // than 64 under one wave -- keep the search per group below.
      const uint32_t* ends = static_cast<const uint32_t*>(seg.aux);
      const uint32_t first = __builtin_amdgcn_readfirstlane(slice.row_begin + wave * 2048);
      const uint32_t end = __builtin_amdgcn_readfirstlane(slice.row_begin + (slice.row_count < wave * 2048 + 2048 ? slice.row_count : wave * 2048 + 2048));
      uint32_t run = __builtin_amdgcn_readfirstlane(run_of_position(ends, seg.aux_size, first));
      uint32_t cursor = first, walked_mask = 0;
      for (uint32_t step = 0; step < 64 && cursor < end; ++step, ++run) {
        const uint32_t run_end = ends[run] + 1 < end ? ends[run] + 1 : end;   // (exclusive)
        if (eval_run(seg, job, run)) {
#pragma unroll
          for (uint32_t k = 0; k < 4; ++k) walked_mask |= rows_in_range8(first + k * 512 + lane * 8, cursor, run_end) << (8 * k);
        }
        cursor = run_end;
      }
      walked = cursor >= end;
      if (walked) mask = walked_mask;
    }
    if (walked) {
    } else if ((job.mode == JOB_SCAN || job.mode == JOB_RANGE) && !(job.flags & JF_NEVER)) {
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
        if (r0 < slice.row_count) {
          const uint32_t valid = (slice.row_count - r0 < 8) ? slice.row_count - r0 : 8;
          mask |= eval8<COMPRESSED>(seg, job, slice.row_begin + r0, valid) << (8 * k);
        }
      }
    } else if (job.mode == JOB_ALL && a.materialize_all) {
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
        if (r0 < slice.row_count) {
          const uint32_t valid = (slice.row_count - r0 < 8) ? slice.row_count - r0 : 8;
          mask |= ((1u << valid) - 1u) << (8 * k);
        }
      }
    }
  }
  return mask;
}


// ---- streaming instantiations (W = 1 | 2 | 4): every segment of the column is a W-byte attribute vector, FoR offset vector
// or int32 value vector.  Lane l of wave w owns, per slice, the four 8-row groups  w*2048 + k*512 + l*8  (k = 0..3).
// A group is always loaded whole (one aligned 8*W-byte access): a group that holds at least one row of the segment lies
// in the same page as that row, so the bytes past the end of the last group are readable; they are masked out below.
template <int W>
struct SliceLoad {
  RawGroup<W> raw[4];
  uint32_t null_byte[4];
  uint32_t bias;          // FrameOfReference: minimum of the wave's 2048-row block (a wave's rows never straddle blocks)
  uint32_t bits;          // W = 16: bits per element of the segment's BitPackingVector
};

template <int W>
__device__ __forceinline__ void issue_loads(SliceLoad<W>& ld, const DevSegment& seg, const ScanJob& job, const Slice& slice, uint32_t wave, uint32_t lane) {
  ld.bias = 0;
  ld.bits = seg_bits(seg);
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) ld.null_byte[k] = 0;
  if (job.mode != JOB_SCAN || (job.flags & JF_NEVER) || slice.row_count == 0) return;
  const bool has_bitmap = seg.nulls != nullptr && seg.encoding != HY_ENC_DICTIONARY;
  const uint32_t last_group = (slice.row_count - 1) & ~7u;
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
    if (slice.row_count != SLICE_ROWS) r0 = r0 < last_group ? r0 : last_group;   // groups past the end re-read the last one
    const uint32_t row = slice.row_begin + r0;
    if constexpr (W == 16) ld.raw[k] = load_packed_group(seg.data, row, seg_bits(seg));
    else ld.raw[k] = load_group<W>(seg.data, row);
    if (has_bitmap) ld.null_byte[k] = as_global<uint8_t>(seg.nulls)[row >> 3];
  }
  if (seg.encoding == HY_ENC_FRAME_OF_REFERENCE) {
    uint32_t r0 = __builtin_amdgcn_readfirstlane(wave) * 2048;
    r0 = r0 < last_group ? r0 : last_group;
    ld.bias = static_cast<uint32_t>(as_global<int32_t>(seg.aux)[(slice.row_begin + r0) / HY_FOR_BLOCK_SIZE]);
  }
}

// Packed 16-bit range test of two rows at once: 1 in a half  <=>  that row is OUTSIDE [lo, lo + span].
// (inline assembly: written with builtins the compiler turns the saturating subtract back into compares + selects)
__device__ __forceinline__ uint32_t pk_outside(uint32_t two_rows, uint32_t lo2, uint32_t span2, uint32_t one2) {
  uint32_t x, t, r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(x) : "v"(two_rows), "v"(lo2));
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(t) : "v"(x), "v"(span2));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(t), "v"(one2));
  return r;
}
// 1 in a half  <=>  that row differs from `value`
__device__ __forceinline__ uint32_t pk_differs(uint32_t two_rows, uint32_t value2, uint32_t one2) {
  uint32_t x, r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(x) : "v"(two_rows), "v"(value2));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(one2));
  return r;
}

// Rows of a group as four dwords of two 16-bit lanes each (row 2i in the low half of dword i).
template <int W>
__device__ __forceinline__ void packed_rows(const RawGroup<W>& g, uint32_t (&d)[4]) {
  if constexpr (W == 2) {
    d[0] = g.v.x; d[1] = g.v.y; d[2] = g.v.z; d[3] = g.v.w;
  } else {
    static_assert(W == 1, "packed evaluation is for 8- and 16-bit elements");
    d[0] = __builtin_amdgcn_perm(g.v.x, g.v.x, 0x0c010c00u); d[1] = __builtin_amdgcn_perm(g.v.x, g.v.x, 0x0c030c02u);
    d[2] = __builtin_amdgcn_perm(g.v.y, g.v.y, 0x0c010c00u); d[3] = __builtin_amdgcn_perm(g.v.y, g.v.y, 0x0c030c02u);
  }
}

// Per-row flags of two groups: bit (8*g + r) <-> row r of group g.  `flag(dword)` returns 0/1 in each 16-bit half.
template <int W, typename Flag>
__device__ __forceinline__ uint32_t two_groups(const RawGroup<W>& g0, const RawGroup<W>& g1, Flag flag) {
  uint32_t d0[4], d1[4];
  packed_rows<W>(g0, d0);
  packed_rows<W>(g1, d1);
  uint32_t acc = 0;   // even rows collect in bits 0..15, odd rows 16 higher
#pragma unroll
  for (int i = 0; i < 4; ++i) acc |= (flag(d0[i]) << (2 * i)) | (flag(d1[i]) << (8 + 2 * i));
  return (acc | (acc >> 15)) & 0xFFFFu;
}

// RANGES: the column has chunks flagged as sorted (their jobs may be JOB_RANGE); the instantiations every other column takes carry none
// of that code.
template <int W, bool RANGES>
__device__ __forceinline__ uint32_t evaluate_loaded(const SliceLoad<W>& ld, const DevSegment& seg, const ScanJob& job, const Slice& slice,
                                                    uint32_t materialize_all, uint32_t wave, uint32_t lane) {
  uint32_t valid = 0xFFFFFFFFu;   // rows of the lane's groups that exist
  if (slice.row_count != SLICE_ROWS) {
    valid = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
      const uint32_t rows = r0 >= slice.row_count ? 0u : (slice.row_count - r0 < 8 ? slice.row_count - r0 : 8u);
      valid |= ((1u << rows) - 1u) << (8 * k);
    }
  }
  if (job.mode == JOB_ALL) return materialize_all ? valid : 0u;
  if (RANGES && job.mode == JOB_RANGE) return job_range_mask(job.lo, job.span, slice.row_begin + wave * 2048 + lane * 8) & valid;   // a sorted chunk: nothing was loaded
  if (job.mode != JOB_SCAN || (job.flags & JF_NEVER)) return 0u;

  const bool invert = job.flags & JF_INVERT;
  const uint32_t null_bits = ld.null_byte[0] | (ld.null_byte[1] << 8) | (ld.null_byte[2] << 16) | (ld.null_byte[3] << 24);   // 0 for dictionaries
  if (job.kind == KIND_NULLTEST) return (invert ? ~null_bits : null_bits) & valid;

  const bool is_dict = seg.encoding == HY_ENC_DICTIONARY;
  uint32_t inside;   // rows whose value lies in [lo, lo + span]
  uint32_t not_null = ~null_bits;
  if constexpr (W == 4 || W == 16) {
    const uint32_t lo = static_cast<uint32_t>(job.lo) - ld.bias, span = static_cast<uint32_t>(job.span);
    inside = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      uint32_t x[8];
      if constexpr (W == 16) unpack_packed_group(ld.raw[k], ld.bits, x);
      else unpack_group<W>(ld.raw[k], x);
      inside |= range_bits8(x, lo, span) << (8 * k);
      if (is_dict && invert) {
#pragma unroll
        for (int j = 0; j < 8; ++j) not_null &= ~((x[j] == job.null_vid ? 1u : 0u) << (8 * k + j));
      }
    }
  } else {
    // The stored elements are < 2^(8W): clip the range to that domain once per wave and compare two rows per
    // instruction.  In terms of the stored element x the predicate is  lo - bias <= x <= lo - bias + span.
    constexpr int64_t max_element = (int64_t{1} << (8 * W)) - 1;
    const int64_t low = static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(job.lo))) - static_cast<int32_t>(ld.bias);
    const int64_t high = low + static_cast<int64_t>(static_cast<uint32_t>(job.span));
    const int64_t clipped_low = low < 0 ? 0 : low, clipped_high = high > max_element ? max_element : high;
    if (clipped_low > clipped_high) {
      inside = 0;
    } else {
      const uint32_t one2 = 0x00010001u;
      const uint32_t lo2 = static_cast<uint32_t>(clipped_low) * 0x00010001u, span2 = static_cast<uint32_t>(clipped_high - clipped_low) * 0x00010001u;
      auto outside = [&](uint32_t d) { return pk_outside(d, lo2, span2, one2); };
      inside = ~(two_groups<W>(ld.raw[0], ld.raw[1], outside) | (two_groups<W>(ld.raw[2], ld.raw[3], outside) << 16));
    }
    if (is_dict && invert && job.null_vid <= static_cast<uint32_t>(max_element)) {   // != : NULL (value id == dictionary size) never matches
      const uint32_t one2 = 0x00010001u, null2 = job.null_vid * 0x00010001u;
      auto differs = [&](uint32_t d) { return pk_differs(d, null2, one2); };
      not_null = two_groups<W>(ld.raw[0], ld.raw[1], differs) | (two_groups<W>(ld.raw[2], ld.raw[3], differs) << 16);
    }
  }
  return (invert ? ~inside : inside) & not_null & valid;
}

// Inclusive prefix sum over the 64 lanes of a wave, six DPP adds (no LDS traffic, unlike __shfl_up's ds_bpermute):
// four row_shr steps scan each 16-lane row (bound_ctrl: lanes without a source add 0), row_bcast:15 / row_bcast:31 carry
// the row totals into the later rows.  Inline assembly because the compiler splits every update_dpp + add into
// v_mov + v_mov_dpp + v_add; the s_nops are the VALU-write -> DPP-read wait states the assembler does not insert.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
  asm volatile(
      "s_nop 4\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(v));
  return v;
}

constexpr uint32_t ROW_PAD = 16;   // slack of a wave's compaction buffer: its output range starts anywhere inside a 128-byte line

__device__ __forceinline__ Slice part_slice(const Part& part, const DevSegment& seg, uint32_t i) {
  Slice slice;
  slice.chunk = part.chunk;
  slice.row_begin = part.first_row + i * SLICE_ROWS;
  slice.row_count = seg.size - slice.row_begin < SLICE_ROWS ? seg.size - slice.row_begin : SLICE_ROWS;
  slice.first_of_chunk = slice.row_begin == 0;
  return slice;
}

// One slice's match mask (bit 8k + j <-> row wave*2048 + k*512 + lane*8 + j) -> RowIDs at the running offset of the chunk's region: the count and
// emit steps of scan_slices (see there), shared with scan_two_columns.  `emitted`: matches of the chunk before this slice, advanced here.
__device__ __forceinline__ void emit_slice_matches(const ScanArgs& a, const Part& part, const Slice& slice, uint32_t mask, uint32_t& emitted, uint32_t& parity, uint32_t* s_wave_count,
                                                   uint16_t* my_rows, uint8_t* my_bytes, uint32_t wave, uint32_t lane) {
  // Transpose the masks inside the wave so that lane L holds the 32 CONSECUTIVE rows [32 L, 32 L + 32) of the wave's
  // 2048: byte k of lane l goes to byte k*64 + l of the wave's 256-byte scratch, lane L reads dword L.  A lane's
  // matches are then one contiguous run of the output, and one prefix sum over the popcounts places them.
  my_bytes[lane] = static_cast<uint8_t>(mask);
  my_bytes[64 + lane] = static_cast<uint8_t>(mask >> 8);
  my_bytes[128 + lane] = static_cast<uint8_t>(mask >> 16);
  my_bytes[192 + lane] = static_cast<uint8_t>(mask >> 24);
  __builtin_amdgcn_wave_barrier();   // LDS operations of one wave execute in order; this only stops reordering
  uint32_t run = reinterpret_cast<const uint32_t*>(my_bytes)[lane];
  __builtin_amdgcn_wave_barrier();
  const uint32_t run_count = __popc(run);
  const uint32_t inclusive = wave_inclusive_scan_u32(run_count);
  const uint32_t my_total = __builtin_amdgcn_readlane(inclusive, 63);
  uint32_t* counts = s_wave_count + parity * 4;
  parity ^= 1;
  if (lane == 0) counts[wave] = my_total;
  __syncthreads();
  const uint32_t c0 = __builtin_amdgcn_readfirstlane(counts[0]), c1 = __builtin_amdgcn_readfirstlane(counts[1]),
                 c2 = __builtin_amdgcn_readfirstlane(counts[2]), c3 = __builtin_amdgcn_readfirstlane(counts[3]);
  const uint32_t my_offset = emitted + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
  emitted += c0 + c1 + c2 + c3;
  if (my_total != 0) {
    // The wave's RowIDs go to elements [first, first + my_total) of the output.  The compaction buffer is laid out so
    // that LDS slot q is output element  line + q,  line = first rounded down to a 128-byte line (16 RowIDs):
    // every store instruction of the body then writes 64 x 16 B = 8 whole, aligned lines.
    const uint64_t first = part.region_base + my_offset;
    const uint32_t skew = static_cast<uint32_t>(first) & 15u;
    const uint32_t end = skew + my_total;
    {
      uint16_t* slot = my_rows + skew + (inclusive - run_count);
      const uint32_t row0 = wave * 2048 + lane * 32;
      while (run) {
        const uint32_t j = __ffs(run) - 1;
        run &= run - 1;
        *slot++ = static_cast<uint16_t>(row0 + j);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (first + my_total > a.capacity) {
      if (lane == 0) *a.overflow = 1;
    } else {
      // RowIDs are written once and not read again by this kernel: nontemporal, 16 bytes (two RowIDs) per lane.
      // Pairs [pair_begin, pair_end) lie completely inside the wave's range; the slot before and the slot after
      // them are written on their own.
      HY_GLOBAL u32x4* out = (HY_GLOBAL u32x4*)(a.matches + (first - skew));
      const uint32_t* pairs = reinterpret_cast<const uint32_t*>(my_rows);
      const uint32_t pair_begin = (skew + 1) / 2, pair_end = end / 2;
      if (a.plain_stores) {
        for (uint32_t q = lane < pair_begin ? lane + 64 : lane; q < pair_end; q += 64) {
          const uint32_t two = pairs[q];
          const u32x4 v = {part.chunk, slice.row_begin + (two & 0xFFFFu), part.chunk, slice.row_begin + (two >> 16)};
          out[q] = v;
        }
      } else {
        for (uint32_t q = lane < pair_begin ? lane + 64 : lane; q < pair_end; q += 64) {
          const uint32_t two = pairs[q];
          const u32x4 v = {part.chunk, slice.row_begin + (two & 0xFFFFu), part.chunk, slice.row_begin + (two >> 16)};
          __builtin_nontemporal_store(v, out + q);
        }
      }
      HY_GLOBAL u32x2* single = (HY_GLOBAL u32x2*)out;
      if (lane == 0 && (skew & 1)) { const u32x2 v = {part.chunk, slice.row_begin + my_rows[skew]}; __builtin_nontemporal_store(v, single + skew); }
      if (lane == 1 && (end & 1)) { const u32x2 v = {part.chunk, slice.row_begin + my_rows[end - 1]}; __builtin_nontemporal_store(v, single + (end - 1)); }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// W = 0: generic instantiation (mixed widths, 64-bit / floating point value segments, reference segments,
// ColumnVsColumn); W = 1 | 2 | 4: streaming instantiation.
//
// One workgroup owns one PART at a time (<= 8 slices = 65 536 rows of ONE chunk; parts b, b + gridDim.x, ...) and walks
// its slices ONCE, in row order.  Per slice (8192 rows; lane l of wave w owns 4 groups of 8 consecutive rows):
//   evaluate  the slice's loads were issued one slice earlier (also across parts), so every lane keeps 8 x 16 B in
//             flight while it evaluates two rows per instruction (packed 16-bit range test); the 32-bit match mask stays
//             in a register
//   count     the masks are transposed inside the wave (256 bytes of LDS) so that lane L holds 32 CONSECUTIVE rows; one
//             DPP prefix scan over the popcounts gives every lane its output position inside the wave and the wave its
//             total; the four wave totals meet in LDS (the only workgroup barrier of the slice)
//   emit      wave w owns rows [w*2048, (w+1)*2048), i.e. a contiguous piece of the output: it compacts its row numbers
//             through its private 4 KiB of LDS and writes RowID pairs with line-aligned, nontemporal 16-byte stores at
//             the running offset of the chunk's output region.
// Reads of slice s+1, the ALU work of slice s and the writes of slice s overlap inside every workgroup, and the
// descriptors of a part (Part, DevSegment, ScanJob) are scalar-loaded one part ahead, off the critical path.
// A Hyrise chunk (<= 65 535 rows) is one part: NO inter-workgroup communication at all.  Only chunks larger than a
// part chain their parts: such a part first counts its matches (a second, cache-resident read of its rows), publishes
// the total as one epoch-tagged 8-byte word (agent-scope relaxed atomic; the word is the flag) and reads the totals of
// the earlier parts of its chunk in parallel (they belong to workgroups that are resident and never wait for later
// parts, so this cannot deadlock as long as the grid is co-resident).
// Every RowID is written exactly once, in (chunk, row) order: bit-identical to the CPU loop's appends.
template <int W, bool RANGES = false>
__global__ __launch_bounds__(256) void scan_slices(const DevSegment* __restrict__ segments_in, const DevSegment* __restrict__ right_in,
                                                   const Slice* __restrict__ slices_in, const ScanJob* __restrict__ jobs_in,
                                                   const Part* __restrict__ parts, ScanArgs a) {
  // The descriptor tables are read-only and never alias the outputs: as __restrict__ kernel parameters their
  // (wave-uniform) loads become scalar loads instead of vector loads with waits.
  a.segments = segments_in;
  a.right = right_in;
  a.slices = slices_in;
  a.jobs = jobs_in;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* s_rows = reinterpret_cast<uint16_t*>(smem);                              // [4 waves][2048 + ROW_PAD] compaction buffers
  uint32_t* s_wave_count = reinterpret_cast<uint32_t*>(smem + (SLICE_ROWS + 4 * ROW_PAD) * 2);   // [2][4] wave totals, double buffered
  uint32_t* s_small = s_wave_count + 8;                                              // [16] reductions
  uint8_t* s_bytes = reinterpret_cast<uint8_t*>(s_small + 16);                       // [4 waves][256] mask transposition

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.trace && tid == 0) a.trace[blockIdx.x * 4 + 0] = wall_clock64();   // (reading HW_ID here with s_getreg makes the register allocator spill)

  constexpr bool GENERIC = W == 0 || W == 8;   // W == 8: the generic instantiation that also reads run-length segments and bit-packed vectors in place
  constexpr int LW = GENERIC ? 1 : W;
  SliceLoad<LW> next;   // streaming state: loads of the next slice to evaluate (possibly of the next part)
  uint32_t part_id = blockIdx.x;
  Part part{};
  DevSegment seg{};
  ScanJob job{};
  if (part_id < a.n_parts) {
    part = parts[part_id];
    seg = a.segments[part.chunk];
    if constexpr (!GENERIC) {
      job = a.jobs[part.chunk];
      issue_loads<W>(next, seg, job, part_slice(part, seg, 0), wave, lane);
    }
  }

  uint32_t parity = 0;
  while (part_id < a.n_parts) {
    const uint32_t next_part_id = part_id + gridDim.x;
    Part next_part = part;
    DevSegment next_seg = seg;
    ScanJob next_job = job;
    if (next_part_id < a.n_parts) {   // one part ahead: these scalar loads complete long before they are needed
      next_part = parts[next_part_id];
      next_seg = a.segments[next_part.chunk];
      if constexpr (!GENERIC) next_job = a.jobs[next_part.chunk];
    }

    // ---- chunks larger than one part: matches in the earlier parts of the same chunk -------------------------------------
    uint32_t before = 0;
    if (part.parts_in_chunk > 1) {
      uint32_t mine = 0;
      for (uint32_t i = 0; i < part.n_slices; ++i) {
        const Slice slice = part_slice(part, seg, i);
        if constexpr (GENERIC) {
          mine += __popc(evaluate_slice<W == 8>(a, slice, seg, wave, lane));
        } else {
          SliceLoad<W> again;
          issue_loads<W>(again, seg, job, slice, wave, lane);
          mine += __popc(evaluate_loaded<W, RANGES>(again, seg, job, slice, a.materialize_all, wave, lane));
        }
      }
      mine = __builtin_amdgcn_readlane(wave_inclusive_scan_u32(mine), 63);
      if (lane == 0) s_small[wave] = mine;
      __syncthreads();
      const uint32_t part_total = s_small[0] + s_small[1] + s_small[2] + s_small[3];
      if (tid == 0) __hip_atomic_store(&a.status[part_id], (static_cast<uint64_t>(a.epoch) << 32) | part_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t i = part.first_part + tid; i < part_id; i += WG_THREADS) {
        uint64_t word;
        do {
          word = __hip_atomic_load(&a.status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (static_cast<uint32_t>(word >> 32) != a.epoch) __builtin_amdgcn_s_sleep(1);
        } while (static_cast<uint32_t>(word >> 32) != a.epoch);
        before += static_cast<uint32_t>(word);
      }
      before = __builtin_amdgcn_readlane(wave_inclusive_scan_u32(before), 63);
      if (lane == 0) s_small[4 + wave] = before;
      __syncthreads();
      before = s_small[4] + s_small[5] + s_small[6] + s_small[7];
    }

    // ---- per-chunk outputs ------------------------------------------------------------------------------------------------
    uint32_t mode = JOB_SCAN;
    if constexpr (!GENERIC) {
      mode = job.mode;
    } else if (!a.right) {
      if (seg.encoding != HY_ENC_REFERENCE) mode = a.jobs[part.chunk].mode;
      else if (seg.ref_chunk_id != 0xFFFFFFFFu && seg.size > 0) mode = a.jobs[seg.ref_chunk_id].mode;
    }
    if (tid == 0 && part.part_in_chunk == 0) {
      a.offsets[part.chunk] = part.region_base;
      if (a.chunk_state) a.chunk_state[part.chunk] = mode == JOB_ALL ? HY_CHUNK_ALL_MATCH : mode == JOB_NONE ? HY_CHUNK_NONE_MATCH : HY_CHUNK_SCANNED;
      if (part.chunk + 1 == a.n_chunks) a.offsets[a.n_chunks] = part.region_base + seg.size;
    }

    // ---- the slices of the part, in row order -----------------------------------------------------------------------------
    uint32_t emitted = before;   // matches of this chunk before the current slice
    uint16_t* my_rows = s_rows + wave * (2048 + ROW_PAD);
    uint8_t* my_bytes = s_bytes + wave * 256;
    uint32_t n_slices = part.n_slices;
    if constexpr (!GENERIC) {
      // A chunk with an early-out (no row can match, or every row matches and all-match chunks own no RowIDs) has nothing to evaluate
      // and nothing to emit: its eight rounds of transposition, scan and barrier were 25 us of a scan whose chunks all take one
      // (IS NULL on a column without NULLs, a clustered table).  Only the next part's first loads are still requested here.
      if (mode == JOB_NONE || (job.flags & JF_NEVER) || (mode == JOB_ALL && !a.materialize_all)) {
        if (next_part_id < a.n_parts) issue_loads<W>(next, next_seg, next_job, part_slice(next_part, next_seg, 0), wave, lane);
        n_slices = 0;
      }
    }
    for (uint32_t i = 0; i < n_slices; ++i) {
      const Slice slice = part_slice(part, seg, i);
      uint32_t mask;
      if constexpr (GENERIC) {
        mask = evaluate_slice<W == 8>(a, slice, seg, wave, lane);
      } else {
        const SliceLoad<W> current = next;
        if (i + 1 < n_slices) issue_loads<W>(next, seg, job, part_slice(part, seg, i + 1), wave, lane);
        else if (next_part_id < a.n_parts) issue_loads<W>(next, next_seg, next_job, part_slice(next_part, next_seg, 0), wave, lane);
        mask = evaluate_loaded<W, RANGES>(current, seg, job, slice, a.materialize_all, wave, lane);
      }
      emit_slice_matches(a, part, slice, mask, emitted, parity, s_wave_count, my_rows, my_bytes, wave, lane);
    }
    if (tid == 0 && part.part_in_chunk + 1 == part.parts_in_chunk && a.counts) a.counts[part.chunk] = mode == JOB_ALL ? seg.size : emitted;
    if (a.trace && tid == 0 && part_id == blockIdx.x) a.trace[blockIdx.x * 4 + 1] = wall_clock64();

    part_id = next_part_id;
    part = next_part;
    seg = next_seg;
    job = next_job;
  }
  if (a.trace && tid == 0) a.trace[blockIdx.x * 4 + 3] = wall_clock64();
}

// ---- ColumnVsColumn as a streaming scan (TPC-H Q4 / Q12: l_commitdate < l_receiptdate) ----------------------------------------------
// What it replaces: ColumnVsColumnTableScanImpl::scan_chunk, column_vs_column_table_scan_impl.cpp:36-187 (typed comparison of the two
// segments' values, NULL on either side never matches).  Both columns are data columns whose every segment is a W-byte attribute /
// offset / value vector (hy_column::stream_width) of one 4-byte type: the kernel is scan_slices' pipeline with two input streams --
// the 16-byte loads of BOTH columns' next slice are in flight while this slice is evaluated -- and the chunk's two dictionaries are
// staged in LDS once per part (dict_words 4-byte entries per side; a dictionary of l_shipdate-like dates is 10 KB), so a row costs two
// LDS reads instead of two global loads (the generic instantiation: 0.29 ms for SF10's 60 M rows, 0.23 of peak).
template <int W, bool IS_FLOAT>
__device__ __forceinline__ uint32_t evaluate_two_loaded(const SliceLoad<W>& l, const SliceLoad<W>& r, const DevSegment& left, const DevSegment& right, const uint32_t* s_left_dictionary,
                                                        const uint32_t* s_right_dictionary, bool staged, uint32_t condition, const Slice& slice, uint32_t wave, uint32_t lane) {
  uint32_t less = 0, equal = 0, greater = 0, nulls = 0, valid = 0;   // bit 8k + j: row j of group k
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t r0 = wave * 2048 + k * 512 + lane * 8;
    const uint32_t rows = r0 >= slice.row_count ? 0u : (slice.row_count - r0 < 8 ? slice.row_count - r0 : 8u);
    valid |= ((1u << rows) - 1u) << (8 * k);
    nulls |= (l.null_byte[k] | r.null_byte[k]) << (8 * k);
    uint32_t x[8], y[8];
    unpack_group<W>(l.raw[k], x);
    unpack_group<W>(r.raw[k], y);
    if (left.encoding == HY_ENC_DICTIONARY) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool is_null = x[j] >= left.aux_size;
        nulls |= (is_null ? 1u : 0u) << (8 * k + j);
        x[j] = staged ? s_left_dictionary[is_null ? 0u : x[j]] : as_global<uint32_t>(left.aux)[is_null ? 0u : x[j]];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += l.bias;
    }
    if (right.encoding == HY_ENC_DICTIONARY) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool is_null = y[j] >= right.aux_size;
        nulls |= (is_null ? 1u : 0u) << (8 * k + j);
        y[j] = staged ? s_right_dictionary[is_null ? 0u : y[j]] : as_global<uint32_t>(right.aux)[is_null ? 0u : y[j]];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] += r.bias;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // the three outcomes of the typed comparison (a NaN: none of them), the condition picks below
      bool lt, eq, gt;
      if constexpr (IS_FLOAT) {
        const float u = __uint_as_float(x[j]), v = __uint_as_float(y[j]);
        lt = u < v; eq = u == v; gt = u > v;
      } else {
        const int32_t u = static_cast<int32_t>(x[j]), v = static_cast<int32_t>(y[j]);
        lt = u < v; eq = u == v; gt = u > v;
      }
      less |= (lt ? 1u : 0u) << (8 * k + j);
      equal |= (eq ? 1u : 0u) << (8 * k + j);
      greater |= (gt ? 1u : 0u) << (8 * k + j);
    }
    __builtin_amdgcn_sched_barrier(0);   // (group by group: all four groups' values in flight at once cost 48 more registers than the lookups' latency is worth)
  }
  uint32_t match;   // like compare<T>: a != b is !(a == b), the orderings are false for unordered operands
  switch (condition) {
    case HY_PRED_EQUALS: match = equal; break;
    case HY_PRED_NOT_EQUALS: match = ~equal; break;
    case HY_PRED_LESS_THAN: match = less; break;
    case HY_PRED_LESS_THAN_EQUALS: match = less | equal; break;
    case HY_PRED_GREATER_THAN: match = greater; break;
    default: match = greater | equal; break;
  }
  return match & ~nulls & valid;
}

// A chunk in which nothing can match: a dictionary segment without entries holds NULLs only (every other shape the host has checked:
// run_scan launches this kernel only for column pairs whose every segment it streams).
__device__ __forceinline__ bool two_columns_never_match(const DevSegment& left, const DevSegment& right) {
  return (left.encoding == HY_ENC_DICTIONARY && left.aux_size == 0) || (right.encoding == HY_ENC_DICTIONARY && right.aux_size == 0);
}

template <int W, bool IS_FLOAT>
__global__ __launch_bounds__(256) void scan_two_columns(const DevSegment* __restrict__ left_in, const DevSegment* __restrict__ right_in, const Part* __restrict__ parts, ScanArgs a,
                                                        uint32_t dict_words) {
  a.segments = left_in;
  a.right = right_in;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* s_rows = reinterpret_cast<uint16_t*>(smem);
  uint32_t* s_wave_count = reinterpret_cast<uint32_t*>(smem + (SLICE_ROWS + 4 * ROW_PAD) * 2);
  uint32_t* s_small = s_wave_count + 8;
  uint8_t* s_bytes = reinterpret_cast<uint8_t*>(s_small + 16);
  uint32_t* s_left_dictionary = reinterpret_cast<uint32_t*>(s_bytes + 4 * 256);   // [dict_words]
  uint32_t* s_right_dictionary = s_left_dictionary + dict_words;                  // [dict_words]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ScanJob scan_everything{};   // (issue_loads asks the job whether the chunk is read at all)
  scan_everything.mode = JOB_SCAN;

  SliceLoad<W> next_left, next_right;
  uint32_t part_id = blockIdx.x;
  if (part_id < a.n_parts) {
    const Part first = parts[part_id];
    const DevSegment left = a.segments[first.chunk], right = a.right[first.chunk];
    if (!two_columns_never_match(left, right)) {
      issue_loads<W>(next_left, left, scan_everything, part_slice(first, left, 0), wave, lane);
      issue_loads<W>(next_right, right, scan_everything, part_slice(first, left, 0), wave, lane);
    }
  }
  uint32_t parity = 0;
  while (part_id < a.n_parts) {
    // (descriptors: scalar loads out of the scalar cache -- the part's own were requested one part earlier, as next_*)
    const Part part = parts[part_id];
    const DevSegment left = a.segments[part.chunk], right = a.right[part.chunk];
    const uint32_t next_part_id = part_id + gridDim.x;
    const Part next_part = parts[next_part_id < a.n_parts ? next_part_id : part_id];
    const DevSegment next_left_seg = a.segments[next_part.chunk], next_right_seg = a.right[next_part.chunk];
    const bool never = two_columns_never_match(left, right);
    const bool next_loads = next_part_id < a.n_parts && !two_columns_never_match(next_left_seg, next_right_seg);
    // the chunk's dictionaries -> LDS (the first barrier also ends the reads of the part before)
    if (dict_words) {
      __syncthreads();
      if (left.encoding == HY_ENC_DICTIONARY) { for (uint32_t i = tid; i < left.aux_size; i += WG_THREADS) s_left_dictionary[i] = as_global<uint32_t>(left.aux)[i]; }
      if (right.encoding == HY_ENC_DICTIONARY) { for (uint32_t i = tid; i < right.aux_size; i += WG_THREADS) s_right_dictionary[i] = as_global<uint32_t>(right.aux)[i]; }
      __syncthreads();
    }
    const uint32_t before = 0;   // (every chunk is one part: run_scan sends tables with larger chunks to the generic instantiation)
    if (tid == 0 && part.part_in_chunk == 0) {
      a.offsets[part.chunk] = part.region_base;
      if (a.chunk_state) a.chunk_state[part.chunk] = HY_CHUNK_SCANNED;
      if (part.chunk + 1 == a.n_chunks) a.offsets[a.n_chunks] = part.region_base + left.size;
    }
    uint32_t emitted = before;
    uint16_t* my_rows = s_rows + wave * (2048 + ROW_PAD);
    uint8_t* my_bytes = s_bytes + wave * 256;
    const uint32_t n_slices = never ? 0u : part.n_slices;
    for (uint32_t i = 0; i < n_slices; ++i) {
      const Slice slice = part_slice(part, left, i);
      const SliceLoad<W> current_left = next_left, current_right = next_right;
      if (i + 1 < n_slices) {
        issue_loads<W>(next_left, left, scan_everything, part_slice(part, left, i + 1), wave, lane);
        issue_loads<W>(next_right, right, scan_everything, part_slice(part, left, i + 1), wave, lane);
      } else if (next_loads) {
        issue_loads<W>(next_left, next_left_seg, scan_everything, part_slice(next_part, next_left_seg, 0), wave, lane);
        issue_loads<W>(next_right, next_right_seg, scan_everything, part_slice(next_part, next_left_seg, 0), wave, lane);
      }
      const uint32_t mask = evaluate_two_loaded<W, IS_FLOAT>(current_left, current_right, left, right, s_left_dictionary, s_right_dictionary, dict_words != 0, a.condition, slice, wave, lane);
      emit_slice_matches(a, part, slice, mask, emitted, parity, s_wave_count, my_rows, my_bytes, wave, lane);
    }
    if (n_slices == 0 && next_loads) {   // (nothing to read here: the next part's first loads are still requested)
      issue_loads<W>(next_left, next_left_seg, scan_everything, part_slice(next_part, next_left_seg, 0), wave, lane);
      issue_loads<W>(next_right, next_right_seg, scan_everything, part_slice(next_part, next_left_seg, 0), wave, lane);
    }
    if (tid == 0 && part.part_in_chunk + 1 == part.parts_in_chunk && a.counts) a.counts[part.chunk] = emitted;
    part_id = next_part_id;
  }
}

// ---- pos lists that reference several chunks --------------------------------------------------------------------------
// The reference splits such a pos list by referenced chunk (split_pos_list_by_chunk_id.cpp:13-59), scans sub-list by
// sub-list and appends the matches in that order (abstract_dereferenced_column_table_scan_impl.cpp:49-106): the chunk's
// output is its matches ordered by (referenced chunk id, position), NULL_ROW_IDs last.  scan_slices produced them in
// position order; one wave per such chunk re-orders its region with a stable LSD radix sort on the referenced chunk id
// (8 bits per pass; a slow path in the reference as well).
__device__ __forceinline__ uint32_t referenced_chunk_of(const hy_row_id* pos_list, uint32_t position, uint32_t n_referenced_chunks) {
  const hy_row_id r = pos_list[position];
  return r.chunk_offset == 0xFFFFFFFFu || r.chunk_id > n_referenced_chunks ? n_referenced_chunks : r.chunk_id;
}

__global__ __launch_bounds__(64) void reorder_by_referenced_chunk(const DevSegment* __restrict__ segments, hy_row_id* matches, const uint64_t* __restrict__ offsets,
                                                                  const uint32_t* __restrict__ counts, hy_row_id* temp, uint32_t n_referenced_chunks) {
  __shared__ uint32_t s_bin[256];
  const uint32_t chunk = blockIdx.x, lane = threadIdx.x;
  const DevSegment seg = segments[chunk];
  if (seg.encoding != HY_ENC_REFERENCE || !seg.data || seg.ref_chunk_id != 0xFFFFFFFFu) return;
  const uint32_t n = counts[chunk];
  if (n < 2) return;
  const hy_row_id* pos_list = static_cast<const hy_row_id*>(seg.data);
  hy_row_id* src = matches + offsets[chunk];
  hy_row_id* dst = temp + offsets[chunk];
  uint32_t passes = 0;
  for (uint32_t shift = 0; shift < 32 && (n_referenced_chunks >> shift) != 0; shift += 8) {
    for (uint32_t b = lane; b < 256; b += 64) s_bin[b] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&s_bin[(referenced_chunk_of(pos_list, src[i].chunk_offset, n_referenced_chunks) >> shift) & 0xFF], 1u);
    __builtin_amdgcn_wave_barrier();
    // exclusive prefix over the 256 bins: four consecutive bins per lane
    const uint32_t b0 = s_bin[4 * lane], b1 = s_bin[4 * lane + 1], b2 = s_bin[4 * lane + 2], b3 = s_bin[4 * lane + 3];
    const uint32_t sum = b0 + b1 + b2 + b3;
    const uint32_t before = wave_inclusive_scan_u32(sum) - sum;
    s_bin[4 * lane] = before;
    s_bin[4 * lane + 1] = before + b0;
    s_bin[4 * lane + 2] = before + b0 + b1;
    s_bin[4 * lane + 3] = before + b0 + b1 + b2;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < n; base += 64) {   // 64 consecutive matches per round keep the order stable
      const uint32_t i = base + lane;
      const bool valid = i < n;
      hy_row_id row{0, 0};
      uint32_t digit = 0;
      if (valid) {
        row = src[i];
        digit = (referenced_chunk_of(pos_list, row.chunk_offset, n_referenced_chunks) >> shift) & 0xFF;
      }
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (uint32_t b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      if (valid) dst[s_bin[digit] + __popcll(peers & ((1ull << lane) - 1))] = row;
      __builtin_amdgcn_wave_barrier();
      if (valid && (peers >> lane) >> 1 == 0) s_bin[digit] += static_cast<uint32_t>(__popcll(peers));   // highest peer advances the bin
      __builtin_amdgcn_wave_barrier();
    }
    hy_row_id* swap = src;
    src = dst;
    dst = swap;
    ++passes;
    __threadfence();   // the next pass reads what this one wrote
  }
  if (passes & 1) {
    for (uint32_t i = lane; i < n; i += 64) dst[i] = src[i];
  }
}

uint64_t* g_trace_buffer = nullptr;
uint32_t g_trace_grid = 0;

__global__ void exclude_chunks(ScanJob* jobs, const uint32_t* excluded, uint32_t n_excluded) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_excluded) jobs[excluded[i]].mode = JOB_NONE;
}

// Host results only: copy every chunk's region into the back-to-back layout (one workgroup per chunk).
__global__ void compact_regions(const hy_row_id* regions, const uint64_t* region_offsets, const uint64_t* dense_offsets, hy_row_id* dense, uint32_t n_chunks) {
  const uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  const uint64_t count = dense_offsets[c + 1] - dense_offsets[c];
  const uint2* src = reinterpret_cast<const uint2*>(regions + region_offsets[c]);
  uint2* dst = reinterpret_cast<uint2*>(dense + dense_offsets[c]);
  for (uint64_t i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
}


// hy_poslist_translate: exclusive prefix sum of the per-chunk match counts (one workgroup; a column has thousands of chunks
// at most) -> where every chunk's PosList begins in the back-to-back output, and the total behind the last one.
__global__ __launch_bounds__(256) void region_prefix(const uint32_t* __restrict__ counts, uint32_t n_chunks, uint64_t* __restrict__ dense_offsets) {
  __shared__ uint64_t s_sum[256];
  const uint32_t per = (n_chunks + 255) / 256;
  const uint32_t begin = min(n_chunks, threadIdx.x * per), end = min(n_chunks, begin + per);
  uint64_t mine = 0;
  for (uint32_t c = begin; c < end; ++c) mine += counts[c];
  s_sum[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t running = 0;
    for (uint32_t t = 0; t < 256; ++t) { const uint64_t v = s_sum[t]; s_sum[t] = running; running += v; }
    dense_offsets[n_chunks] = running;
  }
  __syncthreads();
  uint64_t running = s_sum[threadIdx.x];
  for (uint32_t c = begin; c < end; ++c) { dense_offsets[c] = running; running += counts[c]; }
}

// hy_poslist_translate: chunk c's region -> its place in the back-to-back PosList; a match (c, o) of a scan over
// ReferenceSegments becomes the RowID at position o of chunk c's PosList (table_scan.cpp:158-196), so that the output
// references the data table and never a reference table.  blockIdx.x: chunk, blockIdx.y: quarter of its matches.
__global__ __launch_bounds__(256) void translate_regions(const DevSegment* __restrict__ segments, const hy_row_id* __restrict__ regions, const uint64_t* __restrict__ region_offsets,
                                                         const uint32_t* __restrict__ counts, const uint64_t* __restrict__ dense_offsets, hy_row_id* __restrict__ out,
                                                         uint64_t capacity, uint32_t keep_regions) {
  const uint32_t c = blockIdx.x;
  const uint32_t count = counts[c];
  const uint64_t base = keep_regions ? region_offsets[c] : dense_offsets[c];
  if (base + count > capacity) return;   // the host reports HY_ERR_CAPACITY from the total
  const DevSegment seg = segments[c];
  const uint2* src = reinterpret_cast<const uint2*>(regions + region_offsets[c]);
  uint2* dst = reinterpret_cast<uint2*>(out + base);
  const bool reference = seg.encoding == HY_ENC_REFERENCE;
  const uint2* pos_list = static_cast<const uint2*>(seg.data);
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < count; i += 256 * gridDim.y) {
    uint2 r = src[i];
    if (reference) r = pos_list ? pos_list[r.y] : make_uint2(seg.ref_chunk_id, r.y);
    dst[i] = r;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------

// Launch shape of the persistent scan kernel: min(parts, CUs x resident workgroups per CU).  Workgroups of a chunk
// that spans several parts wait for each other, so the grid must be co-resident: stay one below what the occupancy
// query admits (ROCm 7.2 over-reports by one for SGPR-heavy 256-thread kernels, MI355X_MICROARCH.md "Residency and
// cooperative launch").
using ScanKernel = void (*)(const DevSegment*, const DevSegment*, const Slice*, const ScanJob*, const Part*, ScanArgs);
constexpr size_t SCAN_LDS_BYTES = (SLICE_ROWS + 4 * ROW_PAD) * 2 + 8 * 4 + 16 * 4 + 4 * 256;   // compaction buffers | wave totals | reductions

static uint32_t scan_grid(ScanKernel kernel, uint32_t n_parts) {
  int device = 0, cus = 256;
  (void)hipGetDevice(&device);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  int want = 8;
  if (FIXED_SCAN_WGS_PER_CU > 0) want = static_cast<int>(FIXED_SCAN_WGS_PER_CU);
  int per_cu = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WG_THREADS, SCAN_LDS_BYTES);
  // The over-report only happens where SGPRs are the limiter (API 7-8 blocks per CU); a VGPR- or LDS-limited answer
  // (<= 5) is exact.
  if (per_cu >= 6) per_cu -= 1;
  int use = per_cu < want ? per_cu : want;
  if (use < 1) use = 1;
  const uint32_t resident_max = static_cast<uint32_t>(cus * use);
  return n_parts < resident_max ? (n_parts ? n_parts : 1) : resident_max;
}

static hy_status validate_predicate(const hy_column* column, const hy_predicate* p) {
  const uint32_t c = p->condition;
  const bool null_test = c == HY_PRED_IS_NULL || c == HY_PRED_IS_NOT_NULL;
  const bool like = c >= HY_PRED_LIKE && c <= HY_PRED_NOT_LIKE_INSENSITIVE;
  if (!(c <= HY_PRED_BETWEEN_EXCLUSIVE || null_test || like)) return fail(HY_ERR_UNSUPPORTED, "predicate condition %u stays on the CPU path (In / ExpressionEvaluator)", c);
  const hy_column* data_column = column->is_reference ? column->ref : column;
  if (like) {
    if (!p->match_words || !p->match_word_offsets) return fail(HY_ERR_INVALID, "LIKE scans need the per-chunk dictionary match bitmaps (hy_predicate.match_words)");
    for (uint32_t chunk = 0; data_column && chunk < data_column->n_chunks; ++chunk) {
      if (data_column->host_segments[chunk].encoding != HY_ENC_DICTIONARY) return fail(HY_ERR_UNSUPPORTED, "LIKE on unencoded string segments stays on the CPU path");
    }
    return HY_OK;
  }
  if (!null_test) {
    if (data_column && data_column->has_dictionary_without_values) {
      if (!p->per_chunk_lower || !p->per_chunk_upper) return fail(HY_ERR_INVALID, "string dictionary scan needs per-chunk value ids (hy_predicate.per_chunk_lower/upper)");
    } else if (column->data_type == HY_TYPE_STRING) {
      // dictionary<string> with no dictionary buffers at all (every chunk NULL-only) still needs resolved ids
      if (!p->per_chunk_lower || !p->per_chunk_upper) return fail(HY_ERR_INVALID, "string column scan needs per-chunk value ids");
    } else if (p->value_type != column->data_type) {
      // ColumnVsValueTableScanImpl asserts equal types (column_vs_value_table_scan_impl.cpp:34-36)
      return fail(HY_ERR_INVALID, "literal type %u differs from column type %u: use the lossless predicate cast first", p->value_type, column->data_type);
    }
  }
  return HY_OK;
}

// The column's remembered job table for `pa` (hy_scan_job_cache, hy_device.hpp): *jobs = the table -- prepared here, on `stream`, if the
// predicate is new to the column -- or nullptr when the caller should prepare its own (the table was made on another stream).
static hy_status cached_scan_jobs(const hy_column* column, const PredicateArgs& pa, hipStream_t stream, ScanJob** jobs) {
  *jobs = nullptr;
  hy_scan_job_cache& cache = column->scan_jobs;
  uint8_t key[sizeof(hy_scan_job_cache::Entry::key)];
  std::memset(key, 0, sizeof(key));
  std::memcpy(key, &pa.condition, 4);
  std::memcpy(key + 4, &pa.value_type, 4);
  std::memcpy(key + 8, &pa.value, 8);
  std::memcpy(key + 16, &pa.value2, 8);
  std::memcpy(key + 24, &pa.column_is_nullable, 4);
  std::memcpy(key + 28, &pa.materialize_all, 4);
  std::memcpy(key + 32, &pa.no_ranges, 4);
  std::lock_guard<std::mutex> lock(cache.mutex);
  hy_scan_job_cache::Entry* victim = &cache.entries[0];
  for (hy_scan_job_cache::Entry& entry : cache.entries) {
    if (entry.jobs && std::memcmp(entry.key, key, sizeof(key)) == 0) {
      if (entry.stream != stream) return HY_OK;   // (ordered behind its preparation only on the stream that prepared it)
      entry.used = ++cache.clock;
      *jobs = static_cast<ScanJob*>(entry.jobs);
      return HY_OK;
    }
    if (!entry.jobs ? victim->jobs != nullptr : (victim->jobs && entry.used < victim->used)) victim = &entry;
  }
  // new to the column: the least recently used table is overwritten -- behind, in stream order, every scan of this stream that read it;
  // a table another stream may still be reading is left alone (the caller prepares its own jobs this time)
  if (victim->jobs && victim->stream != stream) return HY_OK;
  if (!victim->jobs) {
    HY_HIP(hipMalloc(&victim->jobs, sizeof(ScanJob) * (size_t{column->n_chunks} + 1)));
    const_cast<hy_column*>(column)->owned.push_back(victim->jobs);   // (under the cache's lock: nothing else appends to a finished column)
  }
  std::memcpy(victim->key, key, sizeof(key));
  victim->stream = stream;
  victim->used = ++cache.clock;
  hipLaunchKernelGGL(prepare_jobs, dim3(column->n_chunks), dim3(256), 0, stream, column->d_segments, column->n_chunks, pa, static_cast<ScanJob*>(victim->jobs), scratch().ticket + 16);
  *jobs = static_cast<ScanJob*>(victim->jobs);
  return HY_OK;
}

size_t scan_jobs_staging_bytes(const hy_column* column, const hy_predicate* predicate) {
  size_t bytes = 4 * 256 + 9 * (size_t{column->n_chunks} + 64);   // prepare_jobs' status word | the per-chunk arrays, 256-byte aligned
  if (predicate->match_words && predicate->match_word_offsets) bytes += 2 * 256 + 8 * (size_t{column->n_chunks} + 2) + 8 * (predicate->match_word_offsets[column->n_chunks] + 2);
  return bytes;
}

// (what run_scan does for its own predicate, for callers that evaluate the jobs themselves)
hy_status prepare_scan_jobs(const hy_column* column, const hy_predicate* predicate, ScanJob* jobs, void* staging) {
  if (column->is_reference || column->is_mvcc) return fail(HY_ERR_INVALID, "prepare_scan_jobs: data columns only");
  HY_TRY(validate_predicate(column, predicate));
  const uint32_t n_chunks = column->n_chunks;
  hipStream_t stream = current_stream();
  PredicateArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.condition = predicate->condition;
  pa.value_type = predicate->value_type;
  pa.value = predicate->value;
  pa.value2 = predicate->value2;
  pa.column_is_nullable = predicate->column_is_nullable;
  pa.no_ranges = 1;   // (fused_rows evaluates JOB_SCAN / JOB_ALL / JOB_NONE)
  char* cursor = static_cast<char*>(staging) + 256;
  auto stage = [&](const void* host, size_t bytes, const void** dev) -> hy_status {
    *dev = nullptr;
    if (!host) return HY_OK;
    if (bytes) HY_HIP(hipMemcpyAsync(cursor, host, bytes, hipMemcpyHostToDevice, stream));
    *dev = cursor;
    cursor += align_up(bytes ? bytes : 4, 256);
    return HY_OK;
  };
  const void* d;
  HY_TRY(stage(predicate->per_chunk_lower, 4 * size_t{n_chunks}, &d)); pa.per_chunk_lower = static_cast<const uint32_t*>(d);
  HY_TRY(stage(predicate->per_chunk_upper, 4 * size_t{n_chunks}, &d)); pa.per_chunk_upper = static_cast<const uint32_t*>(d);
  HY_TRY(stage(predicate->per_chunk_found, size_t{n_chunks}, &d)); pa.per_chunk_found = static_cast<const uint8_t*>(d);
  if (predicate->match_words && predicate->match_word_offsets) {
    const uint64_t total_words = predicate->match_word_offsets[n_chunks];
    HY_TRY(stage(predicate->match_word_offsets, 8 * (size_t{n_chunks} + 1), &d)); pa.match_word_offsets = static_cast<const uint64_t*>(d);
    HY_TRY(stage(predicate->match_words, 8 * (total_words ? total_words : 1), &d)); pa.match_words = static_cast<const uint64_t*>(d);
  }
  if (n_chunks) hipLaunchKernelGGL(prepare_jobs, dim3(n_chunks), dim3(256), 0, stream, column->d_segments, n_chunks, pa, jobs, static_cast<uint32_t*>(staging));
  return HY_OK;
}

hy_status prepare_visibility_scan_jobs(const hy_column* mvcc, uint32_t our_tid, uint32_t snapshot_commit_id, uint32_t can_use_chunk_shortcut, ScanJob* jobs, void* staging) {
  if (!mvcc || !mvcc->is_mvcc || mvcc->is_reference) return fail(HY_ERR_INVALID, "a Validate filter needs the table's column of HY_ENC_MVCC segments");
  const uint32_t n_chunks = mvcc->n_chunks;
  if (n_chunks) hipLaunchKernelGGL(prepare_visibility_jobs, dim3((n_chunks + 255) / 256), dim3(256), 0, current_stream(), mvcc->d_segments, n_chunks, our_tid, snapshot_commit_id, can_use_chunk_shortcut,
                                   jobs, static_cast<uint32_t*>(staging));
  return HY_OK;
}

struct VisibilityArgs {   // hy_validate
  uint32_t our_tid, snapshot, can_use_chunk_shortcut;
};

static hy_status run_scan(const hy_column* column, const hy_column* right, const hy_predicate* predicate, uint32_t condition,
                          const uint32_t* excluded, uint32_t n_excluded, hy_scan_result* result, const VisibilityArgs* visibility = nullptr) {
  // Run-length / bit-packed segments are read in place by a ColumnVsValue / Between / IsNull / Like scan of the data column itself; the
  // two-column scan and the scan through reference segments read the decoded twins (hy_device.hpp) -- a reference segment's `ref`
  // already points at the twin's descriptors, so the jobs are prepared from the twin as well.
  if (right) {
    HY_TRY(plain_column(column, &column));
    HY_TRY(plain_column(right, &right));
  }
  const uint32_t n_chunks = column->n_chunks;
  const hy_column* data_column = column->is_reference ? column->ref : column;
  if (column->is_reference) HY_TRY(plain_column(data_column, &data_column));
  const uint32_t n_data_chunks = data_column ? data_column->n_chunks : 0;
  const bool host_result = result->mem == HY_MEM_HOST;
  if (!result->offsets) return fail(HY_ERR_INVALID, "scan result: offsets array missing");
  if (host_result && (!result->counts || !result->chunk_state)) return fail(HY_ERR_INVALID, "scan result: host results need counts and chunk_state");
  if (!host_result && !(result->flags & HY_SCAN_CHUNK_REGIONS)) return fail(HY_ERR_INVALID, "scan result: device results use the chunk-region layout; set HY_SCAN_CHUNK_REGIONS");
  if (!host_result && result->capacity < column->rows) return fail(HY_ERR_CAPACITY, "scan result: chunk regions need capacity >= %llu rows", static_cast<unsigned long long>(column->rows));
  if (result->capacity && !result->matches) return fail(HY_ERR_INVALID, "scan result: matches buffer missing");
  hipStream_t stream = current_stream();
  Scratch& sc = scratch();

  // Scratch: jobs | per_chunk arrays | excluded | overflow flag | (host results) regions + dense copy + per-chunk arrays
  size_t need = sizeof(ScanJob) * (n_data_chunks + 1) + 3 * 4 * (size_t{n_data_chunks} + 64) + 4 * (size_t{n_excluded} + 64) + 1024;
  const bool like = predicate && predicate->condition >= HY_PRED_LIKE && predicate->condition <= HY_PRED_NOT_LIKE_INSENSITIVE;
  if (like) need += 8 * (size_t{n_data_chunks} + 2) + 8 * (predicate->match_word_offsets[n_data_chunks] + 2) + 1024;
  if (column->multi_chunk_reference) need += sizeof(hy_row_id) * (column->rows + 1) + 256;
  if (host_result) need += 2 * sizeof(hy_row_id) * (column->rows + 1) + 3 * 8 * (size_t{n_chunks} + 2) + 4 * (size_t{n_chunks} + 1) + n_chunks + 8192;
  HY_TRY(sc.reserve(need + 16 * 256));
  ScanJob* d_jobs = carve<ScanJob>(sc, n_data_chunks + 1);
  uint32_t* d_overflow = sc.ticket + 16;   // persistent word, zero unless a scan overflowed (an internal error)
  ScanKernel kernel = !right && column->has_compressed ? scan_slices<8> : scan_slices<0>;
  if (!right && column->stream_width == 16) kernel = column->has_sorted ? scan_slices<16, true> : scan_slices<16>;   // bit-packed vectors of at most 16 bits
  else if (!right && column->stream_width == 1) kernel = column->has_sorted ? scan_slices<1, true> : scan_slices<1>;
  else if (!right && column->stream_width == 2) kernel = column->has_sorted ? scan_slices<2, true> : scan_slices<2>;
  else if (!right && column->stream_width == 4) kernel = column->has_sorted ? scan_slices<4, true> : scan_slices<4>;
  if (like) kernel = column->has_compressed ? scan_slices<8> : scan_slices<0>;   // value-id sets are tested by the generic instantiations

  PredicateArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.materialize_all = (result->flags & HY_SCAN_MATERIALIZE_ALL_MATCH) ? 1 : 0;
  if (visibility && n_data_chunks) {
    hipLaunchKernelGGL(prepare_visibility_jobs, dim3((n_data_chunks + 255) / 256), dim3(256), 0, stream, data_column->d_segments, n_data_chunks, visibility->our_tid,
                       visibility->snapshot, visibility->can_use_chunk_shortcut, d_jobs, d_overflow);
  }
  if (predicate) {
    pa.condition = predicate->condition;
    pa.value_type = predicate->value_type;
    pa.value = predicate->value;
    pa.value2 = predicate->value2;
    pa.column_is_nullable = predicate->column_is_nullable;
    auto stage = [&](const void* host, size_t bytes, const void** dev) -> hy_status {
      *dev = nullptr;
      if (!host) return HY_OK;
      void* d = sc.carve(bytes ? bytes : 4);
      HY_HIP(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream));
      *dev = d;
      return HY_OK;
    };
    const void* d;
    HY_TRY(stage(predicate->per_chunk_lower, 4 * size_t{n_data_chunks}, &d)); pa.per_chunk_lower = static_cast<const uint32_t*>(d);
    HY_TRY(stage(predicate->per_chunk_upper, 4 * size_t{n_data_chunks}, &d)); pa.per_chunk_upper = static_cast<const uint32_t*>(d);
    HY_TRY(stage(predicate->per_chunk_found, size_t{n_data_chunks}, &d)); pa.per_chunk_found = static_cast<const uint8_t*>(d);
    if (predicate->match_words && predicate->match_word_offsets) {
      const uint64_t total_words = predicate->match_word_offsets[n_data_chunks];
      HY_TRY(stage(predicate->match_word_offsets, 8 * (size_t{n_data_chunks} + 1), &d)); pa.match_word_offsets = static_cast<const uint64_t*>(d);
      HY_TRY(stage(predicate->match_words, 8 * (total_words ? total_words : 1), &d)); pa.match_words = static_cast<const uint64_t*>(d);
    }
    // a literal predicate over a data column whose jobs this column remembers (hy_scan_job_cache): no launch
    const bool cacheable = FIXED_SCAN_JOB_CACHE && !visibility && !n_excluded && n_data_chunks && !pa.per_chunk_lower && !pa.per_chunk_upper && !pa.per_chunk_found &&
                           !pa.match_words && !pa.match_word_offsets;
    ScanJob* cached = nullptr;
    if (cacheable) HY_TRY(cached_scan_jobs(data_column, pa, stream, &cached));
    if (cached) {
      d_jobs = cached;
    } else if (n_data_chunks) {
      hipLaunchKernelGGL(prepare_jobs, dim3(n_data_chunks), dim3(256), 0, stream, data_column->d_segments, n_data_chunks, pa, d_jobs, d_overflow);
    }
  }
  if (n_excluded) {
    if (column->is_reference) return fail(HY_ERR_INVALID, "excluded chunks apply to data tables only");
    uint32_t* d_excluded = carve<uint32_t>(sc, n_excluded);
    HY_HIP(hipMemcpyAsync(d_excluded, excluded, 4 * size_t{n_excluded}, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(exclude_chunks, dim3((n_excluded + 255) / 256), dim3(256), 0, stream, d_jobs, d_excluded, n_excluded);
  }

  hy_row_id* d_matches = result->matches;
  uint64_t* d_offsets = result->offsets;
  uint32_t* d_counts = result->counts;
  uint8_t* d_state = result->chunk_state;
  uint64_t device_capacity = result->capacity;
  if (host_result) {
    device_capacity = column->rows;
    d_matches = carve<hy_row_id>(sc, device_capacity + 1);
    d_offsets = carve<uint64_t>(sc, size_t{n_chunks} + 2);
    d_counts = carve<uint32_t>(sc, size_t{n_chunks} + 1);
    d_state = carve<uint8_t>(sc, size_t{n_chunks} + 1);
    if (!d_matches || !d_offsets || !d_counts || !d_state) return fail(HY_ERR_DEVICE, "scratch arena exhausted");
  }

  if (n_chunks == 0) {
    HY_HIP(hipMemsetAsync(d_offsets, 0, 8, stream));
  } else {
    const uint32_t grid = scan_grid(kernel, column->n_parts);
    HY_TRY(sc.begin_launch(column->n_parts, 0));
    ScanArgs a;
    std::memset(&a, 0, sizeof(a));
    a.segments = column->d_segments;
    a.right = right ? right->d_segments : nullptr;
    a.slices = column->d_slices;
    a.jobs = d_jobs;
    a.n_slices = column->n_slices;
    a.n_parts = column->n_parts;
    a.n_chunks = n_chunks;
    a.condition = condition;
    a.materialize_all = pa.materialize_all;
    a.is_null_scan = predicate && predicate->condition == HY_PRED_IS_NULL;
    a.epoch = sc.epoch;
    a.status = sc.status;
    a.matches = d_matches;
    a.capacity = device_capacity;
    a.offsets = d_offsets;
    a.counts = d_counts;
    a.chunk_state = d_state;
    a.overflow = d_overflow;
    a.trace = nullptr;
    // Write-back stores: on this part a 37 : 63 read : write stream runs 13 % faster through the L2 than around it (nontemporal) --
    // tools/hbm_mix.hip shows it for the bare traffic pattern, tools/scan_ab.py for this kernel (profiles/r03_scan_stores.txt).
    a.plain_stores = FIXED_SCAN_NT_STORES ? 0u : 1u;
    if (HY_DEBUG_ENV("HY_SCAN_TRACE")) {
      static uint64_t* trace_buffer = nullptr;
      if (!trace_buffer) (void)hipMalloc(reinterpret_cast<void**>(&trace_buffer), 8 * 4 * 4096);
      a.trace = trace_buffer;
      g_trace_buffer = trace_buffer;
      g_trace_grid = grid;
    }
    hipEvent_t started = nullptr, stopped = nullptr;
    profile_events(&started, &stopped, HY_KERNEL_SCAN);
    // ColumnVsColumn over two data columns of one 4-byte type whose segments are all W-byte vectors: the two-stream kernel, the chunks'
    // dictionaries staged in LDS when the largest pair fits 48 KB (scan_two_columns)
    uint32_t two_width = right && !column->is_reference && !right->is_reference && column->stream_width == right->stream_width && column->data_type == right->data_type &&
                                 (column->data_type == HY_TYPE_INT || column->data_type == HY_TYPE_FLOAT) && !column->has_compressed && !right->has_compressed &&
                                 option(HY_OPT_SCAN_TWO_COLUMNS)
                             ? column->stream_width : 0;
    uint32_t dict_words = 0;
    for (uint32_t c = 0; c < n_chunks && two_width; ++c) {   // every segment: a W-byte vector of 4-byte values (stream_width says W bytes and aligned)
      for (const hy_segment* seg : {&column->host_segments[c], &right->host_segments[c]}) {
        const bool shape = seg->encoding == HY_ENC_UNENCODED ? two_width == 4 : (seg->encoding == HY_ENC_DICTIONARY || seg->encoding == HY_ENC_FRAME_OF_REFERENCE) && seg->width == two_width;
        if (!shape || (seg->data_type != HY_TYPE_INT && seg->data_type != HY_TYPE_FLOAT)) two_width = 0;
        if (seg->encoding == HY_ENC_DICTIONARY) dict_words = std::max(dict_words, seg->aux_size);
      }
    }
    for (uint32_t c = 0; c < n_chunks && two_width; ++c) if (column->host_segments[c].size > PART_SLICES * SLICE_ROWS) two_width = 0;   // (a chunk of several parts)
    if (two_width == 1 || two_width == 2 || two_width == 4) {
      dict_words = (dict_words + 3) & ~3u;
      if (8 * size_t{dict_words} > 48 * 1024) dict_words = 0;   // (dictionaries that large stay in global memory: cache hits at best)
      const size_t lds = SCAN_LDS_BYTES + 8 * size_t{dict_words};
      const bool is_float = column->data_type == HY_TYPE_FLOAT;
      int device = 0, cus = 256;
      (void)hipGetDevice(&device);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      auto launch = [&](auto kernel_of_shape) -> hy_status {
        int per_cu = 0;
        HY_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel_of_shape), WG_THREADS, lds));
        const uint32_t resident = static_cast<uint32_t>(cus) * static_cast<uint32_t>(std::max(1, std::min(per_cu, 8)));
        hipExtLaunchKernelGGL(kernel_of_shape, dim3(std::max(1u, std::min(column->n_parts, resident))), dim3(WG_THREADS), lds, stream, started, stopped, 0, a.segments, a.right,
                              column->d_parts, a, dict_words);
        return HY_OK;
      };
      if (two_width == 1) HY_TRY(is_float ? launch(scan_two_columns<1, true>) : launch(scan_two_columns<1, false>));
      else if (two_width == 2) HY_TRY(is_float ? launch(scan_two_columns<2, true>) : launch(scan_two_columns<2, false>));
      else HY_TRY(is_float ? launch(scan_two_columns<4, true>) : launch(scan_two_columns<4, false>));
    } else
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(WG_THREADS), SCAN_LDS_BYTES, stream, started, stopped, 0, a.segments, a.right, a.slices, a.jobs,
                          column->d_parts, a);
  }
  if (column->multi_chunk_reference && predicate && n_chunks && d_counts) {   // (Validate keeps position order, validate.cpp:236-253)
    hy_row_id* d_temp = carve<hy_row_id>(sc, column->rows + 1);
    if (!d_temp) return fail(HY_ERR_DEVICE, "scratch arena exhausted");
    hipLaunchKernelGGL(reorder_by_referenced_chunk, dim3(n_chunks), dim3(64), 0, stream, column->d_segments, d_matches, d_offsets, d_counts, d_temp, n_data_chunks);
  }
  HY_HIP(hipGetLastError());

  if (host_result) {
    uint32_t overflow = 0;
    HY_HIP(hipMemcpyAsync(result->counts, d_counts, 4 * size_t{n_chunks}, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipMemcpyAsync(result->chunk_state, d_state, size_t{n_chunks}, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipMemcpyAsync(&overflow, d_overflow, 4, hipMemcpyDeviceToHost, stream));
    HY_HIP(hipStreamSynchronize(stream));
    if (overflow) {
      (void)hipMemsetAsync(d_overflow, 0, 4, stream);
      return fail(HY_ERR_DEVICE, "scan overflowed its own region buffer (internal error)");
    }
    // back-to-back layout for the host: offsets[c+1] - offsets[c] = RowIDs written for chunk c
    uint64_t total = 0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
      result->offsets[c] = total;
      const bool elided = result->chunk_state[c] == HY_CHUNK_ALL_MATCH && !pa.materialize_all;
      total += elided ? 0 : result->counts[c];
    }
    result->offsets[n_chunks] = total;
    result->total_matches = total;
    if (total > result->capacity) return fail(HY_ERR_CAPACITY, "scan produced %llu RowIDs, capacity is %llu", static_cast<unsigned long long>(total), static_cast<unsigned long long>(result->capacity));
    if (total) {
      uint64_t* d_dense_offsets = carve<uint64_t>(sc, size_t{n_chunks} + 2);
      hy_row_id* d_dense = carve<hy_row_id>(sc, total);
      if (!d_dense_offsets || !d_dense) return fail(HY_ERR_DEVICE, "scratch arena exhausted");
      HY_HIP(hipMemcpyAsync(d_dense_offsets, result->offsets, 8 * (size_t{n_chunks} + 1), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(compact_regions, dim3(n_chunks), dim3(256), 0, stream, d_matches, d_offsets, d_dense_offsets, d_dense, n_chunks);
      HY_HIP(hipMemcpyAsync(result->matches, d_dense, sizeof(hy_row_id) * total, hipMemcpyDeviceToHost, stream));
      HY_HIP(hipStreamSynchronize(stream));
    }
  } else {
    result->total_matches = 0;
  }
  return HY_OK;
}

// hy_poslist_translate's kernels, queued on the current stream (hy_device.hpp): dense_offsets[n_chunks] = the total, in device memory.
hy_status poslist_translate_queued(const hy_column* scanned, const hy_scan_result* result, uint32_t layout, hy_row_id* out, uint64_t capacity, uint64_t* dense_offsets) {
  const uint32_t n_chunks = scanned->n_chunks;
  hipStream_t stream = current_stream();
  hipLaunchKernelGGL(region_prefix, dim3(1), dim3(256), 0, stream, result->counts, n_chunks, dense_offsets);
  hipLaunchKernelGGL(translate_regions, dim3(n_chunks, 4), dim3(256), 0, stream, scanned->d_segments, result->matches, result->offsets, result->counts, dense_offsets, out,
                     capacity, layout == HY_POSLIST_CHUNK_REGIONS ? 1u : 0u);
  HY_HIP(hipGetLastError());
  return HY_OK;
}

}  // namespace hy

using namespace hy;

extern "C" {

// debug only (HY_SCAN_TRACE): copies the per-workgroup phase stamps of the last scan; not part of the public header
int hy_debug_scan_trace(uint64_t* out, uint32_t capacity_wgs) {
  if (!g_trace_buffer) return 0;
  const uint32_t n = g_trace_grid < capacity_wgs ? g_trace_grid : capacity_wgs;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(out, g_trace_buffer, size_t{n} * 32, hipMemcpyDeviceToHost);
  return static_cast<int>(n);
}

hy_status hy_table_scan(const hy_column* column, const hy_predicate* predicate, const uint32_t* excluded_chunks,
                        uint32_t n_excluded, hy_scan_result* result) {
  if (!column || !predicate || !result) return fail(HY_ERR_INVALID, "hy_table_scan: null argument");
  HY_TRY(on_this_device(column, "hy_table_scan"));
  if (n_excluded && !excluded_chunks) return fail(HY_ERR_INVALID, "hy_table_scan: excluded chunk list missing");
  for (uint32_t i = 0; i < n_excluded; ++i) {
    if (excluded_chunks[i] >= column->n_chunks) return fail(HY_ERR_INVALID, "excluded chunk id %u out of range", excluded_chunks[i]);
  }
  if (column->is_mvcc || (column->ref && column->ref->is_mvcc)) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
  HY_TRY(validate_predicate(column, predicate));
  if (column->multi_chunk_reference && result->mem == HY_MEM_DEVICE && !result->counts) {
    return fail(HY_ERR_INVALID, "scans of pos lists that span several chunks need result->counts (their matches are re-ordered by referenced chunk)");
  }
  return run_scan(column, nullptr, predicate, 0, excluded_chunks, n_excluded, result);
}

hy_status hy_validate(const hy_column* mvcc, uint32_t our_tid, uint32_t snapshot_commit_id, uint32_t can_use_chunk_shortcut, hy_scan_result* result) {
  if (!mvcc || !result) return fail(HY_ERR_INVALID, "hy_validate: null argument");
  const hy_column* data = mvcc->is_reference ? mvcc->ref : mvcc;
  if (!data || !data->is_mvcc) return fail(HY_ERR_INVALID, "hy_validate needs a column of HY_ENC_MVCC segments (or reference segments into one)");
  const VisibilityArgs visibility{our_tid, snapshot_commit_id, can_use_chunk_shortcut};
  return run_scan(mvcc, nullptr, nullptr, 0, nullptr, 0, result, &visibility);
}

hy_status hy_table_scan_columns(const hy_column* left, const hy_column* right, uint32_t condition,
                                hy_scan_result* result) {
  if (!left || !right || !result) return fail(HY_ERR_INVALID, "hy_table_scan_columns: null argument");
  HY_TRY(on_this_device(left, "hy_table_scan_columns"));
  HY_TRY(on_this_device(right, "hy_table_scan_columns"));
  if (left->is_mvcc || right->is_mvcc || (left->ref && left->ref->is_mvcc) || (right->ref && right->ref->is_mvcc)) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
  if (condition > HY_PRED_GREATER_THAN_EQUALS) return fail(HY_ERR_INVALID, "ColumnVsColumn supports binary comparisons only");
  if (left->n_chunks != right->n_chunks) return fail(HY_ERR_INVALID, "columns of one table must have the same chunk count");
  for (uint32_t c = 0; c < left->n_chunks; ++c) {
    if (left->host_segments[c].size != right->host_segments[c].size) return fail(HY_ERR_INVALID, "chunk %u: segment sizes differ", c);
  }
  if (left->data_type == HY_TYPE_STRING || right->data_type == HY_TYPE_STRING) return fail(HY_ERR_UNSUPPORTED, "string ColumnVsColumn scans stay on the CPU path");
  return run_scan(left, right, nullptr, condition, nullptr, 0, result);
}

hy_status hy_poslist_translate(const hy_column* scanned, const hy_scan_result* result, uint32_t layout, hy_row_id* out, uint64_t capacity, uint64_t* n_out) {
  if (!scanned || !result || !n_out) return fail(HY_ERR_INVALID, "hy_poslist_translate: null argument");
  if (layout != HY_POSLIST_DENSE && layout != HY_POSLIST_CHUNK_REGIONS) return fail(HY_ERR_INVALID, "hy_poslist_translate: unknown layout %u", layout);
  if (layout == HY_POSLIST_CHUNK_REGIONS && capacity < scanned->rows) return fail(HY_ERR_CAPACITY, "hy_poslist_translate: chunk regions need capacity >= %llu rows", static_cast<unsigned long long>(scanned->rows));
  *n_out = 0;
  if (result->mem != HY_MEM_DEVICE) return fail(HY_ERR_INVALID, "hy_poslist_translate reads a device-memory scan result (host results are back to back already)");
  if (!(result->flags & HY_SCAN_CHUNK_REGIONS) || !(result->flags & HY_SCAN_MATERIALIZE_ALL_MATCH)) {
    return fail(HY_ERR_INVALID, "hy_poslist_translate: the scan must have run with HY_SCAN_CHUNK_REGIONS | HY_SCAN_MATERIALIZE_ALL_MATCH");
  }
  if (!result->matches || !result->offsets || !result->counts) return fail(HY_ERR_INVALID, "hy_poslist_translate: matches, offsets and counts of the scan result are needed");
  if (capacity && !out) return fail(HY_ERR_INVALID, "hy_poslist_translate: output buffer missing");
  const uint32_t n_chunks = scanned->n_chunks;
  if (n_chunks == 0) return HY_OK;
  hipStream_t stream = current_stream();
  Scratch& sc = scratch();
  HY_TRY(sc.reserve(8 * (size_t{n_chunks} + 2) + 4096));
  uint64_t* d_dense_offsets = carve<uint64_t>(sc, size_t{n_chunks} + 2);
  if (!d_dense_offsets) return fail(HY_ERR_DEVICE, "scratch arena exhausted");
  HY_TRY(poslist_translate_queued(scanned, result, layout, out, capacity, d_dense_offsets));
  uint64_t total = 0;
  HY_HIP(hipMemcpyAsync(&total, d_dense_offsets + n_chunks, 8, hipMemcpyDeviceToHost, stream));
  HY_HIP(hipStreamSynchronize(stream));
  *n_out = total;
  if (layout == HY_POSLIST_DENSE && total > capacity) return fail(HY_ERR_CAPACITY, "hy_poslist_translate: %llu RowIDs, capacity is %llu", static_cast<unsigned long long>(total), static_cast<unsigned long long>(capacity));
  return HY_OK;
}

}  // extern "C"
