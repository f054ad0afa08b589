/*
 * hyrise_amd.h -- C ABI of the MI355X-native execution hot path for Hyrise.
 *
 * This is the drop-in boundary: the three read-only operators whose `_on_execute()` bodies
 * (reference: src/lib/operators/abstract_read_only_operator.hpp:20-22) are replaced call these
 * entry points with plain pointers and sizes.  No C++ types, no torch types, no exceptions cross
 * this line.  Every function returns an `hy_status`; a non-zero status has a message retrievable
 * through `hy_last_error()` (the Hyrise adapter turns it into `Fail(...)`, i.e. std::logic_error,
 * reference: src/lib/utils/assert.hpp:48-70).  `HY_ERR_UNSUPPORTED` means "shape not handled on
 * the device; run the stock CPU operator" and is never an execution failure.
 *
 * Layout contract (all restated from the reference):
 *   RowID            {u32 chunk_id, u32 chunk_offset}, NULL_ROW_ID = {~0u, ~0u}   types.hpp:97-117,150
 *   ValueSegment<T>  T values[n] (+ optional null bitmap)                        value_segment.hpp:84-85
 *   DictionarySegment<T>  sorted unique T dictionary[d]; attribute vector of value ids as
 *                    u8/u16/u32 (FixedWidthInteger); NULL == value id d           dictionary_segment.hpp:88-90
 *   FrameOfReferenceSegment<int32>  i32 block_minima[ceil(n/2048)], offsets as u8/u16/u32,
 *                    optional null bitmap; value = (i32)offset + minimum           frame_of_reference_segment.hpp:49,94-97
 *   null bitmap      libstdc++ vector<bool> words: bit (i % 64) of u64 word (i / 64)
 *   ReferenceSegment pos list of RowIDs into the chunks of another column          reference_segment.hpp:20-48
 */
#ifndef HYRISE_AMD_H_
#define HYRISE_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HY_ABI_VERSION 4   /* 2: hy_segment carries sorted_by and bits; 3: hy_join_result flags / status (HY_JOIN_ASYNC), hy_set_option;
                            * 4: hy_result_pool_*, hy_poslist_gather (device-resident PosLists behind _on_execute()), 15 options (were 33) */

typedef int32_t hy_status;
enum {
  HY_OK = 0,
  HY_ERR_INVALID = 1,     /* malformed argument: the adapter raises std::logic_error            */
  HY_ERR_UNSUPPORTED = 2, /* shape not handled on the device: adapter runs the stock CPU path    */
  HY_ERR_DEVICE = 3,      /* HIP runtime failure (message carries the hipError string)            */
  HY_ERR_CAPACITY = 4     /* caller-provided output buffer too small                              */
};

/* Mirrors hyrise::DataType (all_type_variant.hpp:34-39,52). */
enum { HY_TYPE_NULL = 0, HY_TYPE_INT = 1, HY_TYPE_LONG = 2, HY_TYPE_FLOAT = 3, HY_TYPE_DOUBLE = 4, HY_TYPE_STRING = 5 };

/* Mirrors hyrise::PredicateCondition, same numeric order (types.hpp:160-179). */
enum {
  HY_PRED_EQUALS = 0, HY_PRED_NOT_EQUALS = 1, HY_PRED_LESS_THAN = 2, HY_PRED_LESS_THAN_EQUALS = 3,
  HY_PRED_GREATER_THAN = 4, HY_PRED_GREATER_THAN_EQUALS = 5,
  HY_PRED_BETWEEN_INCLUSIVE = 6, HY_PRED_BETWEEN_LOWER_EXCLUSIVE = 7, HY_PRED_BETWEEN_UPPER_EXCLUSIVE = 8,
  HY_PRED_BETWEEN_EXCLUSIVE = 9,
  HY_PRED_IN = 10, HY_PRED_NOT_IN = 11, HY_PRED_LIKE = 12, HY_PRED_NOT_LIKE = 13, HY_PRED_LIKE_INSENSITIVE = 14,
  HY_PRED_NOT_LIKE_INSENSITIVE = 15, HY_PRED_IS_NULL = 16, HY_PRED_IS_NOT_NULL = 17
};

/* Mirrors hyrise::JoinMode, same numeric order (types.hpp:210). */
enum {
  HY_JOIN_INNER = 0, HY_JOIN_LEFT = 1, HY_JOIN_RIGHT = 2, HY_JOIN_FULL_OUTER = 3, HY_JOIN_CROSS = 4, HY_JOIN_SEMI = 5,
  HY_JOIN_ANTI_NULL_AS_TRUE = 6, HY_JOIN_ANTI_NULL_AS_FALSE = 7
};

/* Aggregate functions handled by AggregateHash (window_function_traits.hpp:11-77). */
enum {
  HY_AGG_MIN = 0, HY_AGG_MAX = 1, HY_AGG_SUM = 2, HY_AGG_AVG = 3, HY_AGG_COUNT = 4, HY_AGG_COUNT_DISTINCT = 5,
  HY_AGG_STDDEV_SAMP = 6, HY_AGG_ANY = 7
};

enum { HY_ENC_UNENCODED = 0, HY_ENC_DICTIONARY = 1, HY_ENC_FRAME_OF_REFERENCE = 2, HY_ENC_REFERENCE = 3,
       HY_ENC_MVCC = 4 /* not a column encoding: a chunk's MvccData, see hy_validate */,
       HY_ENC_RUN_LENGTH = 5 /* RunLengthSegment<T> (run_length_segment.hpp): data = run values (T[aux_size]), aux = inclusive end
                              * positions (uint32_t[aux_size]), nulls = per-run NULL flags as BYTES (uint8_t[aux_size], cast), width =
                              * sizeof(T).  The runs stay runs in device memory: hy_table_scan reads them in place (one search of
                              * the end positions per eight rows, run_length_segment_iterable.hpp:100-160); the other operators read a
                              * ValueSegment twin that a device kernel decodes from them the first time one of them asks. */ };

/* HY_ENC_LZ4 = 6: an LZ4Segment<T> of a numeric type (lz4_segment.hpp:25-60, lz4_segment.cpp:125-226): `data` points at ONE hy_lz4_blocks
 * (host memory: such a segment is handed over with HY_MEM_HOST), width = sizeof(T), nulls = the optional null bitmap like a ValueSegment's.
 * The blocks -- compressed one by one with LZ4_compress_*_usingDict against the segment's dictionary, so that each decompresses on its own --
 * travel to the device as they are and a kernel decompresses them there (one wavefront per block, the block's output staged in LDS); the
 * column then holds the ValueSegment the segment was compressed from (the reference, too, decompresses an LZ4Segment to scan it).
 * LZ4 string segments stay with the host like every string payload. */
enum { HY_ENC_LZ4 = 6 };
typedef struct hy_lz4_blocks {
  const void* const* blocks;      /* [block_count] the compressed blocks (pmr_vector<pmr_vector<char>>: one buffer each)             */
  const uint32_t* block_bytes;    /* [block_count] their compressed sizes                                                           */
  uint32_t block_count;
  uint32_t block_size;            /* decompressed bytes of every block but the last (LZ4Encoder::BLOCK_SIZE = 16 384), at most 65 536  */
  uint32_t last_block_size;       /* ... of the last                                                                                */
  uint32_t dictionary_bytes;      /* 0: a single block, compressed without a dictionary                                            */
  const void* dictionary;
} hy_lz4_blocks;

/* hy_segment::sorted_by: the chunk is individually sorted by this column (Chunk::individually_sorted_by, chunk.hpp:160-176;
 * HY_SORT_* = hyrise::SortMode + 1, types.hpp:219).  ColumnVsValue / ColumnBetween scans then find the matching row range with
 * binary searches instead of reading the segment (sorted_segment_search.hpp:20-384, column_vs_value_table_scan_impl.cpp:46-55). */
enum { HY_SORT_NONE = 0, HY_SORT_ASCENDING_NULLS_FIRST = 1, HY_SORT_DESCENDING_NULLS_FIRST = 2, HY_SORT_ASCENDING_NULLS_LAST = 3,
       HY_SORT_DESCENDING_NULLS_LAST = 4 };

/* Where the pointers of a descriptor / result live. */
enum { HY_MEM_HOST = 0, HY_MEM_DEVICE = 1 };

#define HY_INVALID_VALUE_ID 0xFFFFFFFFu /* "not found" (types.hpp:152), NOT the NULL value id */
#define HY_FOR_BLOCK_SIZE 2048u         /* frame_of_reference_segment.hpp:49 */
#define HY_INVALID_COLUMN 0xFFFFu       /* COUNT(*) argument (INVALID_COLUMN_ID)                      */

typedef struct hy_row_id {
  uint32_t chunk_id;
  uint32_t chunk_offset;
} hy_row_id;

typedef struct hy_column hy_column; /* opaque: one column of a table, chunk by chunk, resident on the device */

/*
 * One segment (= one column of one chunk).  Plain views of what Hyrise keeps in pmr_vectors.
 *   UNENCODED          data = T values[size]                       width = sizeof(T)
 *   DICTIONARY         data = attribute vector (value ids)          width = 1|2|4 (FixedWidthInteger), or width = 0 and
 *                      bits = 1..32: a BitPackingVector (bitpacking_vector_type.hpp:18) -- compact::vector<uint32_t, 0, uint64_t>:
 *                      element i occupies bits [i * bits, (i + 1) * bits) of a little-endian stream of 64-bit words.  It stays
 *                      packed in device memory: hy_table_scan unpacks in registers (bitpacking_decompressor.hpp:35-37), the other
 *                      operators read a FixedWidthInteger twin unpacked on the device on first use
 *                      aux  = T dictionary[aux_size] sorted unique; may be NULL for HY_TYPE_STRING
 *                      (then comparisons must be passed as pre-resolved value ids, see hy_predicate)
 *   FRAME_OF_REFERENCE data = offset values                         width = 1|2|4, or bit-packed (width = 0, bits) like DICTIONARY
 *                      aux  = int32 block_minima[aux_size]
 *   REFERENCE          data = hy_row_id pos_list[size]  (NULL => EntireChunkPosList{ref_chunk_id,size})
 *                      ref  = the referenced column (data segments only; reference_segment.hpp:36-38)
 * nulls: optional bitmap (see header comment); NULL when the segment stores no NULLs.
 * Dictionary segments carry NULLs as value id == aux_size instead (dictionary_segment.cpp:139-141).
 */
typedef struct hy_segment {
  uint32_t encoding;   /* HY_ENC_* */
  uint32_t data_type;  /* HY_TYPE_* of the logical column */
  uint32_t size;       /* rows in this chunk */
  uint32_t width;      /* bytes per element of `data` */
  const void* data;
  const void* aux;
  uint32_t aux_size;
  uint32_t ref_chunk_id;     /* REFERENCE: common chunk id if the pos list references a single chunk, else ~0u */
  const uint64_t* nulls;
  const hy_column* ref;      /* REFERENCE only */
  uint32_t sorted_by;        /* HY_SORT_*: 0 unless the chunk is flagged as sorted by this column */
  uint32_t bits;             /* bits per element of a BitPackingVector in `data` (width == 0), else 0 */
} hy_segment;

/* A literal as the scan implementations receive it after the lossless cast (table_scan.cpp:312-452). */
typedef union hy_value {
  int32_t i32;
  int64_t i64;
  float f32;
  double f64;
  uint32_t value_id;
} hy_value;

/*
 * Predicate of ColumnVsValue / ColumnBetween / ColumnIsNull scans.
 *   condition   HY_PRED_*; two-sided (between) conditions use value and value2.
 *   value_type  HY_TYPE_* of value/value2; must equal the column type (ColumnVsValueTableScanImpl asserts this,
 *               column_vs_value_table_scan_impl.cpp:34-36).
 *   Dictionary segments whose dictionary is not on the device (strings): the caller resolves the literal per chunk
 *   exactly as column_vs_value_table_scan_impl.cpp:211-226 / column_between_table_scan_impl.cpp:112-124 do and passes
 *   per_chunk_lower / per_chunk_upper (value ids, HY_INVALID_VALUE_ID = past the end) plus, for =/!=,
 *   per_chunk_found (1 if dictionary[lower] == literal).  Arrays have one entry per chunk of the column.
 */
typedef struct hy_predicate {
  uint32_t condition;
  uint32_t value_type;
  hy_value value;
  hy_value value2;
  const uint32_t* per_chunk_lower;
  const uint32_t* per_chunk_upper;
  const uint8_t* per_chunk_found;
  uint32_t column_is_nullable; /* Table::column_is_nullable(); matters for dictionary "matches all" early-outs */
  uint32_t reserved;
  /* HY_PRED_LIKE / NOT_LIKE / LIKE_INSENSITIVE / NOT_LIKE_INSENSITIVE on dictionary (incl. FixedString) segments
   * (ColumnLikeTableScanImpl::_scan_dictionary_segment, column_like_table_scan_impl.cpp:69-140): the adapter runs Hyrise's
   * LikeMatcher over every chunk's dictionary (`_find_matches_in_dictionary`, small host work) and passes the result as
   * bitmaps: bit v of the words starting at match_words[match_word_offsets[c]] says whether value id v of data chunk c
   * satisfies the predicate (the matcher already applies NOT); match_word_offsets has one entry per data chunk plus the
   * end.  The device scans the attribute vectors against the bitmaps; the NULL value id never matches. */
  const uint64_t* match_words;
  const uint64_t* match_word_offsets;
} hy_predicate;

/* Per-chunk classification the scan reports (TableScan::PerformanceData counters, table_scan.hpp:56-69). */
enum {
  HY_CHUNK_SCANNED = 0,      /* PosList materialised in `matches`                                              */
  HY_CHUNK_ALL_MATCH = 1,    /* every row matches: count == size, NO RowIDs written (EntireChunkPosList)        */
  HY_CHUNK_NONE_MATCH = 2    /* early-out, count == 0                                                           */
};

/*
 * Result of a scan.  `matches` holds the PosLists of all chunks back to back, in input-chunk order; chunk c owns
 * matches[offsets[c] .. offsets[c+1]).  Positions are relative to the *input* chunk (scan_chunk() contract,
 * abstract_dereferenced_column_table_scan_impl.cpp:75-80), ascending, bit-identical to the CPU operator's PosList.
 * All arrays are caller-allocated, in the memory space `mem`.
 */
typedef struct hy_scan_result {
  uint32_t mem;            /* HY_MEM_HOST | HY_MEM_DEVICE */
  uint32_t flags;          /* HY_SCAN_* */
  hy_row_id* matches;      /* capacity RowIDs */
  uint64_t capacity;
  uint64_t* offsets;       /* [n_chunks + 1] */
  uint32_t* counts;        /* [n_chunks] matches per chunk (== size for ALL_MATCH chunks) */
  uint8_t* chunk_state;    /* [n_chunks] HY_CHUNK_* */
  uint64_t total_matches;  /* filled for HY_MEM_HOST results; for device results read offsets[n_chunks] */
} hy_scan_result;

#define HY_SCAN_MATERIALIZE_ALL_MATCH 1u /* also write RowIDs for ALL_MATCH chunks (what the CPU impl does internally) */
/* Layout of `matches`.  Host results are always back to back: offsets[c+1] == offsets[c] + RowIDs written for chunk c.
 * Device results use CHUNK REGIONS (the flag must be set to acknowledge it): chunk c's PosList starts at
 * offsets[c] = number of rows in all chunks before c and holds counts[c] RowIDs (none for an elided ALL_MATCH chunk);
 * offsets[n_chunks] = total rows; capacity must be >= total rows.  The kernel then needs no global prefix sum at all. */
#define HY_SCAN_CHUNK_REGIONS 2u

/* ---- runtime -------------------------------------------------------------------------------------------------- */
int32_t hy_abi_version(void);
hy_status hy_init(int32_t device);            /* selects the HIP device of the process (one process per GPU): every thread that calls
                                               * into the library afterwards is bound to it on its first call */
hy_status hy_shutdown(void);
const char* hy_last_error(void);              /* thread-local message of the last non-OK status */
hy_status hy_set_stream(void* hip_stream);    /* thread-local stream used for all launches; NULL = default stream     */
hy_status hy_synchronize(void);
hy_status hy_device_malloc(void** ptr, size_t bytes);
hy_status hy_device_free(void* ptr);
hy_status hy_memcpy_h2d(void* dst, const void* src, size_t bytes);
hy_status hy_memcpy_d2h(void* dst, const void* src, size_t bytes);
hy_status hy_device_count(int32_t* count);
/* Measurement hook: when enabled (> 0), operator calls time their dominant kernel (scan_slices, join probe, aggregate
 * accumulate) with a pair of HIP events on the launch stream; enabled = n > 1 times every n-th call only (a timed
 * launch costs several microseconds of stream time).  hy_profile_read() waits for the recorded events and returns the
 * summed elapsed time and the number of timed launches since profiling was (re-)enabled. */
hy_status hy_set_profiling(int32_t enabled);
hy_status hy_profile_read(float* total_milliseconds, uint32_t* launches);
/* The same per kernel, without resetting: an operator chain (bench.py's TableScan + JoinHash step) reads the time of each of its
 * timed kernels separately before hy_profile_read() clears the session. */
enum { HY_KERNEL_OTHER = 0, HY_KERNEL_SCAN = 1, HY_KERNEL_JOIN_PROBE = 2, HY_KERNEL_JOIN_COUNT = 3, HY_KERNEL_JOIN_BUILD = 4, HY_KERNEL_AGGREGATE = 5,
       HY_KERNEL_PROJECTION = 6, HY_KERNEL_KINDS = 8 };
hy_status hy_profile_read_kernel(uint32_t kernel, float* total_milliseconds, uint32_t* launches);
/* What such an event pair measures beyond the kernel itself (the pair is stamped from the dispatch packet): the elapsed time of an
 * empty kernel, median of 32 launches, measured once per process.  A profiler's per-kernel duration is shorter by about this much. */
hy_status hy_profile_event_overhead(float* milliseconds);

/* ---- result-buffer pool: device memory for PosLists that outlive the call that wrote them -----------------------------------------
 * The reference's operators hand their results on as tables whose ReferenceSegments share PosLists (AbstractPosList,
 * storage/pos_lists/abstract_pos_list.hpp:18-75; reference_segment.hpp:36-38; table_scan.cpp:207-210; join_output_writing.cpp:95-200): a PosList
 * lives as long as a table references it.  An adapter that keeps operator chains in HBM (DevicePosList in hyrise_amd/host/hyrise_host.hpp,
 * INTEGRATION.md section 3) takes the memory of such PosLists from this pool and gives it back when the PosList object dies; blocks are
 * handed out again in stream order (the next owner's stream waits for what the last owner's stream still had queued), hipMalloc runs only when
 * nothing fits.  Process-wide per device, thread-safe.
 *   hy_result_pool_acquire       one buffer of at least `bytes` bytes (a scan's PosLists, a gathered PosList)
 *   hy_result_pool_acquire_pair  the two PosLists of a join with `rows` RowIDs each: every list an allocation of its own, the first starting on
 *                                a 2 MiB boundary, the second 1.25 MiB past one -- two streams written at the same index at the same time then
 *                                use different memory channels; prefers the pair a calibration kept
 *   hy_result_pool_release       either kind, one pointer at a time (NULL: nothing)
 *   hy_result_pool_calibrate     where a join's output lists lie in HBM decides the emit kernel's speed by up to 20 %, reproducibly per allocation
 *                                (profiles/r04_join_placement.txt, r05_placement_probe.txt): `candidates` fresh pairs for `rows` RowIDs are
 *                                allocated, the join left x right (mode) runs a few times into each (HIP events on the calling thread's stream),
 *                                the fastest pair stays in the pool as the one hy_result_pool_acquire_pair hands out first (with
 *                                HY_POOL_KEEP_MEDIAN the median candidate stays too, second in line -- what an uncalibrated pool gets on
 *                                average), the others are freed.  ms_per_candidate ([candidates], may be NULL): milliseconds per join;
 *                                *chosen: the fastest.  Run it once, outside any timed region, when the process knows its largest join.
 *   hy_result_pool_trim          frees every buffer of the calling thread's device that nobody holds (waits for the thread's stream) */
#define HY_POOL_KEEP_MEDIAN 1u
hy_status hy_result_pool_acquire(uint64_t bytes, void** ptr);
hy_status hy_result_pool_acquire_pair(uint64_t rows, hy_row_id** left, hy_row_id** right);
hy_status hy_result_pool_release(void* ptr);
hy_status hy_result_pool_calibrate(const hy_column* left, const hy_column* right, uint32_t mode, uint64_t rows, uint32_t candidates, uint32_t flags,
                                   float* ms_per_candidate, uint32_t* chosen);
hy_status hy_result_pool_trim(void);
hy_status hy_result_pool_stats(uint64_t* held_bytes, uint64_t* in_use_bytes, uint32_t* calibrated_pairs);

/* ---- options: which of several equivalent paths / launch shapes an operator takes ---------------------------------------------
 * Process-wide integers with the defaults below.  EVERY setting produces the same results: the tests force each path with them and
 * compare it with the oracle, the tools time one path against another in one process.  The library reads no environment variable
 * (a build with -DHY_DEBUG_SWITCHES adds the trace / timing / parts-switched-off aids of DESIGN.md section 6; release builds do not
 * contain them). */
enum {
  HY_OPT_ALLOW_ANY_ARCH = 0,         /* 0    hy_init accepts a device that is not gfx950 (nothing is tuned for it)                    */
  HY_OPT_JOIN_RANK_TABLE = 1,        /* 1    unique dense-ish integer build keys get a rank table (else: the sorted directory)          */
  HY_OPT_JOIN_HINT = 2,              /* 1    later joins over a resident build column fill the table in one checked pass              */
  HY_OPT_JOIN_BREAK_HINT = 3,        /* 0    tests: hand the checked fill a hint that does not hold (the join must notice and rerun)   */
  HY_OPT_JOIN_PKFK = 4,              /* 1    the primary-key / foreign-key probe kernels (join_pkfk.hpp)                              */
  HY_OPT_JOIN_LDS_BUILD = 5,         /* 1    rank tables below 2^20 key values are staged in LDS ...                                  */
  HY_OPT_JOIN_LDS_BUILD_TILES = 6,   /* 2048 ... from this many probe tiles on                                                        */
  HY_OPT_JOIN_FILL_WGS_PER_CU = 7,   /* 4    the checked one-pass fill wave by wave (rank_table_fill_waves: this many resident workgroups per
                                      *      CU, <= 8); 0 = one short-lived workgroup per slice (rank_table_fill_checked)              */
  HY_OPT_JOIN_HAND_OVER_RANKS = 8,   /* 2^20 Inner PK-FK joins whose probe keys have no locality, whose probe side has at least this many rows and whose
                                      *      rank table has a megabyte or more: pass 1 leaves the partners' ranks behind (4 bytes a probe row), pass 2
                                      *      reads them back instead of looking every key up again; 0 = never, 1 = whenever the keys lack locality */
  HY_OPT_AGG_PARTITION_BITS = 9,     /* 0    partition bits of the many-groups path, 0 = derived from the row count; > 0 also forces the path */
  HY_OPT_AGG_SPILL_SHIFT = 10,       /* 3    aggregate_rows gives up when rows >> shift left its LDS tables                           */
  HY_OPT_AGG_SMALL_DOMAIN = 11,      /* 1    sd_groups / sd_wide (aggregate_small.hpp) for the shapes they accept                     */
  HY_OPT_FUSED_SMALL_DOMAIN = 12,    /* 1    fused_small_domain for the shapes it accepts                                             */
  HY_OPT_SCAN_TWO_COLUMNS = 13,      /* 1    ColumnVsColumn over two data columns of W-byte vectors: the two-stream kernel with the chunks' dictionaries in LDS */
  HY_OPT_STAR_FUSED_PROBE = 14,      /* 1    hy_star_join_aggregate probes every dimension in ONE pass over the fact table (csrc/join_star.hpp) where the
                                      *      shape allows; 0 = one hy_join_hash per dimension                                                      */
  HY_OPT_STAR_FUSED_FINISH = 15,     /* 1    ... and groups the survivors where it finds them (star_finish) where every GROUP BY column and aggregate input
                                      *      is an int / long column: no RowIDs, exports, projection, hy_aggregate_hash; 0 = those four steps     */
  HY_OPT_COUNT = 16
  /* (rounds 4-5 had 33: launch shapes and store flavours whose A/B timings were flat twice are constants now -- csrc/hy_options.hpp --,
   *  the radix-partitioned join path, which lost to the rank table read in place by 1.7 x, is gone) */
};
hy_status hy_set_option(uint32_t option, int64_t value);
hy_status hy_get_option(uint32_t option, int64_t* value);

/* ---- several GPUs in one process, and the collectives between them (csrc/comm.hip: RCCL over xGMI) ----------------------------
 * Hyrise is one process whose operators run on scheduler workers (scheduler/operator_task.cpp:163-200): the multi-GPU shape is one
 * worker thread per GPU.  hy_bind_device makes the CALLING thread work on `device` (its stream, pools and every column it creates
 * belong to that device; threads that never call it use hy_init's device).  hy_comm_init_all creates one communicator per device of
 * the process (RCCL's single-process mode, ncclCommInitAll); hy_comm_init_rank one per process (id from hy_comm_unique_id, carried to
 * the other processes by the caller).  The collectives run on the calling thread's stream like a kernel launch (hy_synchronize waits
 * for them).  A thread that drives several devices' communicators brackets each collective for all of them with hy_comm_group_begin /
 * hy_comm_group_end.  What the sharded operators exchange: fixed-slot partial aggregates (all_reduce), (key, partial)
 * tables and build-side columns (all_gather), (key, RowID) tuples by key % G (all_to_all_v: grouped ncclSend / ncclRecv).
 * A device list that names ONE device n times (worker threads that share a GPU; RCCL refuses such a list) gets communicators that
 * exchange through that GPU's memory: same entry points, same results. */
typedef struct hy_comm hy_comm;
enum { HY_COMM_SUM = 0, HY_COMM_MIN = 1, HY_COMM_MAX = 2 };
enum { HY_COMM_ID_BYTES = 128 };
hy_status hy_bind_device(int32_t device);
hy_status hy_comm_unique_id(void* id_128_bytes);
hy_status hy_comm_init_rank(const void* id_128_bytes, uint32_t world, uint32_t rank, hy_comm** comm);
hy_status hy_comm_init_all(const int32_t* devices, uint32_t n_devices, hy_comm** comms /* [n_devices] */);
hy_status hy_comm_destroy(hy_comm* comm);
hy_status hy_comm_rank(const hy_comm* comm, uint32_t* rank, uint32_t* world);
hy_status hy_comm_group_begin(void);
hy_status hy_comm_group_end(void);
hy_status hy_comm_all_reduce(hy_comm* comm, const void* send, void* recv, uint64_t count, uint32_t data_type /* HY_TYPE_INT / LONG / FLOAT / DOUBLE */, uint32_t op);
hy_status hy_comm_all_gather(hy_comm* comm, const void* send, void* recv /* world x bytes_per_rank */, uint64_t bytes_per_rank);
/* send / recv: the bytes for / from rank 0, rank 1, ... back to back */
hy_status hy_comm_all_to_all_v(hy_comm* comm, const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes);

/* ---- residency cache: a column made device-visible once (encoded segments are immutable,
 *      abstract_encoded_segment.hpp:12-17) ------------------------------------------------------------------------ */
/* mem == HY_MEM_HOST: segment buffers are copied to HBM and owned by the hy_column.
 * mem == HY_MEM_DEVICE: pointers are device pointers and stay owned by the caller; the call does not wait for the device: the column's
 *                       descriptor tables are complete in the order of the calling thread's stream (every entry point this thread calls
 *                       afterwards sees them; another thread synchronises with this one first -- hy_synchronize -- like for the data
 *                       the pointers name). */
hy_status hy_column_create(const hy_segment* segments, uint32_t n_chunks, uint32_t mem, hy_column** out);
hy_status hy_column_destroy(hy_column* column);
hy_status hy_column_row_count(const hy_column* column, uint64_t* rows);
hy_status hy_column_chunk_count(const hy_column* column, uint32_t* chunks);

/* ---- TableScan (replaces TableScan::_on_execute's per-chunk scan_chunk() fan-out, table_scan.cpp:97-240) --------- */
/* ColumnVsValue / ColumnBetween / ColumnIsNull (column_vs_value_table_scan_impl.cpp, column_between_table_scan_impl.cpp,
 * column_is_null_table_scan_impl.cpp), over data or reference columns.  excluded_chunks may be NULL. */
hy_status hy_table_scan(const hy_column* column, const hy_predicate* predicate, const uint32_t* excluded_chunks,
                        uint32_t n_excluded, hy_scan_result* result);
/* ColumnVsColumn (column_vs_column_table_scan_impl.cpp:36-187): left <condition> right, both columns of one table. */
hy_status hy_table_scan_columns(const hy_column* left, const hy_column* right, uint32_t condition,
                                hy_scan_result* result);

/* ---- Validate (replaces Validate::_on_execute / _validate_chunks, validate.cpp:32-55,87-314) ----
 * The MVCC visibility filter that sits in front of every TableScan of an SQL-driven plan.  The table's MvccData
 * (storage/mvcc_data.hpp) is passed as one "column" of HY_ENC_MVCC segments, one per chunk:
 *     data  = transaction ids  (uint32_t[size])       aux   = begin commit ids (uint32_t[size])
 *     nulls = END commit ids   (uint32_t[size], cast) aux_size = the chunk's max_begin_cid
 *     ref_chunk_id = the chunk's invalid_row_count, bit 31 set while the chunk is still mutable
 *     width = 4, data_type = HY_TYPE_INT
 * or, for a reference table, as a column of HY_ENC_REFERENCE segments whose `ref` is such an MVCC column.
 * A row is visible iff  snapshot < end_cid && ((snapshot >= begin_cid) != (row_tid == our_tid))   (validate.cpp:47-55).
 * Chunks that are entirely visible (immutable, snapshot >= max_begin_cid, no invalid rows, and the transaction has no
 * in-flight Delete: can_use_chunk_shortcut, validate.cpp:57-68,116-127) are reported as HY_CHUNK_ALL_MATCH (the adapter
 * emits an EntireChunkPosList / reuses the input pos list, validate.cpp:207-211,282-284).  The result has the layout of
 * hy_table_scan: positions inside the input chunks, in position order. */
hy_status hy_validate(const hy_column* mvcc, uint32_t our_tid, uint32_t snapshot_commit_id, uint32_t can_use_chunk_shortcut,
                      hy_scan_result* result);

/* The literal handling TableScan::create_impl performs before it picks a scan implementation (table_scan.cpp:336-366 for
 * `column OP literal`, :406-448 for `column BETWEEN a AND b`): each literal is cast to the column's type without loss
 * (lossless_cast.hpp:32-187) or, where that is impossible but an equivalent predicate exists, the predicate is adjusted --
 * `float_column < 3.1` becomes `float_column <= 3.0999999f` (lossless_predicate_cast.hpp:23-64, .cpp:14-73).  Pure host
 * arithmetic, no device needed.  literal2 is NULL for the binary conditions.  On HY_OK out->condition / value_type / value /
 * value2 are what hy_table_scan takes (the other fields are zeroed); HY_ERR_UNSUPPORTED where the reference falls back to its
 * ExpressionEvaluator scan (`float_column = 3.1`, a NULL literal, a bound that does not fit the column type).  String literals
 * are outside hy_value: against a string column they need no cast, against anything else there is none. */
hy_status hy_predicate_cast(uint32_t condition, uint32_t column_type, uint32_t literal_type, const hy_value* literal, uint32_t literal2_type,
                            const hy_value* literal2, hy_predicate* out);

/* The PosLists a scan hands to the next operator, WITHOUT leaving device memory (replaces the output assembly of
 * TableScan::_on_execute, table_scan.cpp:158-196: the matches of a scan over ReferenceSegments are translated through the
 * input PosList, so the output always references the data table, never another reference table).
 * `result`: a HY_MEM_DEVICE result of hy_table_scan / hy_table_scan_columns / hy_validate over `scanned`, produced with
 * HY_SCAN_CHUNK_REGIONS | HY_SCAN_MATERIALIZE_ALL_MATCH.  Over a reference column a match (c, o) is replaced by the RowID at
 * position o of chunk c's PosList ((ref_chunk_id, o) for an entire-chunk PosList), NULL RowIDs included as they are; over a
 * data column the RowIDs are copied.  `out` is device memory with room for `capacity` RowIDs:
 *   HY_POSLIST_DENSE          ONE back-to-back PosList, the chunks' matches in chunk order.  HY_ERR_CAPACITY with *n_out = the
 *                             needed capacity if `out` is too small.
 *   HY_POSLIST_CHUNK_REGIONS  one PosList per input chunk, chunk c's at out + result->offsets[c] (result->counts[c] RowIDs): the
 *                             reference's output shape -- an output chunk per input chunk, and a PosList that referenced one
 *                             chunk still does (guarantee_single_chunk, table_scan.cpp:176-178).  capacity >= the column's rows.
 * *n_out (host): RowIDs written in total -- the one value that crosses to the host (8 bytes; the call waits for the stream). */
enum { HY_POSLIST_DENSE = 0, HY_POSLIST_CHUNK_REGIONS = 1 };
hy_status hy_poslist_translate(const hy_column* scanned, const hy_scan_result* result, uint32_t layout, hy_row_id* out, uint64_t capacity, uint64_t* n_out);

/* The RowIDs a join hands on when an input was a reference table (write_output_segments, join_output_writing.cpp:95-200: the positions a
 * join finds in a reference table are dereferenced through the input's PosLists, so the output references the data table), WITHOUT leaving
 * device memory: out[i] = the RowID at position positions[i] = (chunk c, offset o) of `reference` -- a column of HY_ENC_REFERENCE segments
 * (any column of the input table that shares the PosLists in question) -- i.e. pos_list(c)[o], (ref_chunk_id, o) for an entire-chunk
 * PosList; a NULL position (outer joins) stays NULL_ROW_ID.  positions / out: device memory, n RowIDs each; queued on the calling thread's
 * stream. */
hy_status hy_poslist_gather(const hy_column* reference, const hy_row_id* positions, uint64_t n, hy_row_id* out);

/* ---- Projection arithmetic (the ArithmeticExpressions a Projection evaluates through the
 * ExpressionEvaluator, operators/projection.cpp + expression/evaluation/expression_functors.hpp:127-213) -------------------
 * result = left <op> right, element-wise over a table's rows; an operand is a column of the table (data or reference
 * segments, any supported encoding) or a literal.  Same numeric order as hyrise::ArithmeticOperator
 * (expression/arithmetic_expression.hpp).  Semantics of the reference:
 *   result type  expression_common_type (expression_utils.cpp:172-204): Double if either is Double; Long with Float -> Double;
 *                Long with Int -> Long; Float if either is Float; else Int
 *   + - *        computed in std::common_type_t of the operand types, then cast to the result type; NULL if an operand is NULL
 *   /            NULL if an operand is NULL or the divisor is 0; computed in the result type (integers truncate)
 *   %            NULL as for /; `%` for two integral operands, std::fmod otherwise
 * The result is a new device-resident column of unencoded value segments (result type, chunked like the column operand(s),
 * with a null vector) that the following operator -- typically hy_aggregate_hash: Q1/Q6 never materialise the expression
 * on the host -- takes like any other column; hy_column_read_chunk copies a chunk back.  Destroy it with hy_column_destroy. */
enum { HY_ARITH_ADD = 0, HY_ARITH_SUB = 1, HY_ARITH_MUL = 2, HY_ARITH_DIV = 3, HY_ARITH_MOD = 4 };
typedef struct hy_operand {
  const hy_column* column;   /* NULL: the operand is `literal` */
  uint32_t literal_type;     /* HY_TYPE_INT / LONG / FLOAT / DOUBLE, or HY_TYPE_NULL for a NULL literal */
  hy_value literal;
} hy_operand;
hy_status hy_projection_arithmetic(uint32_t op, const hy_operand* left, const hy_operand* right, hy_column** result);
/* Copies one chunk of an unencoded device-resident column to host memory: values[rows] of the column's type and, if
 * null_words is not NULL, the null bitmap (bit i of word i/64; rows/64 rounded up words). */
hy_status hy_column_read_chunk(const hy_column* column, uint32_t chunk, void* values, uint64_t* null_words);
uint32_t hy_column_data_type(const hy_column* column);
uint32_t hy_column_chunk_rows(const hy_column* column, uint32_t chunk);

/* ---- JoinHash (replaces JoinHash::_on_execute, join_hash.cpp:116-225,270-572) ----------------------------------- */
typedef struct hy_join_result {
  uint32_t mem;
  uint32_t radix_bits;            /* in: 0xFFFFFFFF = derive like JoinHash::calculate_radix_bits; out: value used      */
  hy_row_id* left_pos;            /* [capacity] RowIDs into the LEFT input                                             */
  hy_row_id* right_pos;           /* [capacity] RowIDs into the RIGHT input (Semi/Anti*: unused, may be NULL -- the     *
                                   *  probe side of those modes is the LEFT input and only left_pos is written)       */
  uint64_t capacity;
  uint64_t* slice_offsets;        /* [slice_capacity + 1] boundaries of the per-probe-slice PosLists                   */
  uint32_t slice_capacity;
  uint32_t n_slices;              /* out */
  uint64_t n_pairs;               /* out */
  uint32_t left_is_build;         /* out: JoinHash::PerformanceData::left_input_is_build_side                          */
  uint32_t flags;                 /* in: HY_JOIN_ASYNC or 0                                                            */
  struct hy_join_status* status;  /* in: HY_JOIN_ASYNC: device memory the join's last planning kernel writes; else unused */
} hy_join_result;

/* HY_JOIN_ASYNC (mem == HY_MEM_DEVICE only): the call returns as soon as the join's kernels are queued on the calling thread's
 * stream -- no host round trip, so a chain (the next scan, the next join) is queued while this one runs.  What the host normally
 * learns -- the pair count, the number of output PosLists, whether they fit the capacities (a join that does not fit writes
 * nothing), whether the build column still is what its key hint said (hy_join_hash below; if not, nothing is written either) --
 * goes to `status` in DEVICE memory, where a following kernel can read it; n_pairs / n_slices of the result are NOT valid until
 * hy_join_hash_finish has run.  Shapes whose build side needs a host decision (the first join over a column, duplicate or unsorted
 * build keys) run synchronously as always and fill `status` at the end: the flag is a permission, not a promise. */
#define HY_JOIN_ASYNC 1u
typedef struct hy_join_status {
  uint64_t n_pairs;
  uint32_t n_slices;
  uint32_t fits;                  /* 1: n_pairs <= capacity and n_slices <= slice_capacity, the PosLists are written        */
  uint32_t build_confirmed;       /* 0: the build column contradicted its key hint -- hy_join_hash_finish runs the join again */
  uint32_t error;                 /* a probe row with more than 4 194 303 partners (HY_ERR_UNSUPPORTED)                      */
  uint64_t reserved;
} hy_join_status;

/* Equi-join of two numeric columns (int32 / int64 / float / double, any two: both sides are cast to JoinHashTraits'
 * HashedType first, join_hash_traits.hpp:15-40 -- the larger integer type, the larger floating type, or THE floating
 * type of an integer x floating join -- and compared there; NaN keys find nothing).  String columns: HY_ERR_UNSUPPORTED
 * here -- the adapter joins them as DictionarySegment<int64> views whose dictionaries hold one id per distinct string,
 * unique << 20 | std::hash(string) & 0xFFFFF, which this join partitions and filters exactly like the strings
 * (INTEGRATION.md section 3).
 * Pair order == the CPU operator's concatenated probe() output (join_hash_steps.hpp:624-792): by radix partition
 * (std::hash<HashedType> of the key -- the identity for integers, libstdc++'s for float / double), then probe row, then
 * build-side insertion order.
 * A result that does not fit `capacity` / `slice_capacity` is HY_ERR_CAPACITY: nothing is written to the PosList or
 * slice_offsets buffers (device-memory results included -- the check happens on the device between the two probe passes),
 * n_pairs and n_slices report what the join needs.  With mem = HY_MEM_DEVICE the PosLists stay in HBM for the next
 * operator and the call makes no host round trip between the passes. */
hy_status hy_join_hash(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result);
/* Completes a HY_JOIN_ASYNC join (same arguments): waits for the calling thread's stream, reads result->status, fills n_pairs /
 * n_slices and returns what the synchronous call would have returned (HY_ERR_CAPACITY with the needed sizes; a build column that
 * contradicted its hint: the hint is dropped and the join runs again, synchronously, before this call returns).  A no-op for a
 * result without the flag. */
hy_status hy_join_hash_finish(const hy_column* left, const hy_column* right, uint32_t mode, hy_join_result* result);

/* A secondary join predicate  left_column <condition> right_column  (OperatorJoinPredicate, operator_join_predicate.hpp;
 * evaluated like MultiPredicateJoinEvaluator::satisfies_all_predicates, multi_predicate_join_evaluator.hpp:44-54, on every
 * pair the primary equality finds: join_hash_steps.hpp:727-747, 869-876).  The columns are columns of the join's LEFT and
 * RIGHT input table (same chunk layout as the respective key column), numeric, of any two types: the values are compared
 * in their common C++ type like the reference's comparator functors; a NULL on either side fails the predicate. */
typedef struct hy_join_predicate {
  const hy_column* left_column;
  const hy_column* right_column;
  uint32_t condition;             /* HY_PRED_EQUALS .. HY_PRED_GREATER_THAN_EQUALS */
  uint32_t reserved;
} hy_join_predicate;
#define HY_MAX_SECONDARY_PREDICATES 4
/* hy_join_hash with secondary predicates (n_secondary = 0: the same join).  HY_JOIN_ANTI_NULL_AS_TRUE with secondary
 * predicates is HY_ERR_UNSUPPORTED, as in JoinHash::supports (join_hash.cpp:39-44). */
hy_status hy_join_hash_predicates(const hy_column* left, const hy_column* right, uint32_t mode, const hy_join_predicate* secondary,
                                  uint32_t n_secondary, hy_join_result* result);
hy_status hy_join_hash_radix_bits(uint64_t build_rows, uint64_t probe_rows, uint32_t* radix_bits);
/* Upper bound for result->capacity without running the join (Semi/Anti: probe rows; others: exact pair count). */
hy_status hy_join_hash_count(const hy_column* left, const hy_column* right, uint32_t mode, uint64_t* n_pairs);

/* write_output_chunks' chunking of a join result (join_output_writing.cpp:245-296; JoinHash always allows the merge,
 * join_hash.cpp:563): one output chunk per non-empty PosList of slice_offsets[0 .. n_slices], after merging a PosList of fewer
 * than 1000 pairs with its successors while the sum stays below 4000.  chunk_offsets (room for n_slices + 1 values) receives
 * the pair ranges of the output chunks, *n_chunks their number.  Host arrays; pure host arithmetic. */
hy_status hy_join_output_chunks(const uint64_t* slice_offsets, uint32_t n_slices, uint64_t* chunk_offsets, uint32_t* n_chunks);

/* ---- AggregateHash (replaces AggregateHash::_on_execute, aggregate_hash.cpp:1180-1372) -------------------------- */
typedef struct hy_aggregate_spec {
  uint32_t function;              /* HY_AGG_* */
  const hy_column* column;        /* NULL for COUNT(*) */
} hy_aggregate_spec;

typedef struct hy_aggregate_column {
  uint32_t data_type;             /* out: HY_TYPE_* of the result column (window_function_traits.hpp)                 */
  uint32_t reserved;
  void* values;                   /* [group_capacity] int64 / double / int32 / float                                   */
  uint8_t* is_null;               /* [group_capacity] 1 = NULL (a group that saw only NULLs)                           */
} hy_aggregate_column;

typedef struct hy_aggregate_result {
  uint32_t mem;                   /* HY_MEM_HOST, or HY_MEM_DEVICE: group_row_ids / values / is_null are device buffers (the
                                   * groups are ordered on the host either way; a device result is uploaded once, so that an
                                   * operator chain can end on the device)                                              */
  uint32_t group_capacity;
  uint32_t n_groups;              /* out */
  uint32_t reserved;
  hy_row_id* group_row_ids;       /* [group_capacity] representative row per group (write_groupby_output)              */
  hy_aggregate_column* columns;   /* [n_aggregates] */
} hy_aggregate_result;

/* Group order == the CPU operator's: first occurrence, or ascending key under the immediate-key shortcut
 * (aggregate_hash.cpp:388-401, 770-804).  Up to sixteen GROUP BY columns (the reference: any number, aggregate_hash.cpp:1184-1198); more
 * answer HY_ERR_UNSUPPORTED. */
hy_status hy_aggregate_hash(const hy_column* const* groupby_columns, uint32_t n_groupby,
                            const hy_aggregate_spec* aggregates, uint32_t n_aggregates, hy_aggregate_result* result);

/* ---- fused TableScan(s) -> Projection -> AggregateHash over ONE data table --------------------
 * The plan shape of TPC-H Q1 / Q6 (tpch_queries.cpp:60-80, 206-210): a chain of ColumnVsValue / Between / IsNull / LIKE scans
 * on columns of one table (table_scan.cpp:97-240), a Projection of arithmetic expressions over the survivors
 * (operators/projection.cpp, expression_functors.hpp:127-213) and an AggregateHash of them (aggregate_hash.cpp:1180-1372).
 * The reference materialises a PosList per scan and an ExpressionResult vector per expression node; hy_table_scan ->
 * hy_poslist_translate -> hy_projection_arithmetic -> hy_aggregate_hash do the same in HBM.  This entry point evaluates the
 * whole chain in ONE pass over the table: a row is tested against every filter (the same normalised per-chunk tests
 * hy_table_scan evaluates: NULL never matches), the expressions are computed for the rows that pass -- cell by cell with
 * the types and NULL rules of hy_projection_arithmetic -- and fed to the accumulators of hy_aggregate_hash.  No PosList and
 * no expression column is written.  Results are those of the chain: same groups, same group order, same values (sums
 * of floating-point inputs within the tolerance hy_aggregate_hash states); group_row_ids name rows of the DATA table (the
 * chain's name rows of its last intermediate table).
 * Expressions are in postfix order over at most HY_MAX_EXPRESSION_NODES nodes and three stack slots:
 * l_extendedprice * (1 - l_discount) is  COLUMN l_extendedprice, LITERAL 1, COLUMN l_discount, ARITHMETIC SUB, ARITHMETIC MUL.
 * Validate as a filter (the plan shape of a SQL pipeline: GetTable -> Validate -> scans -> Projection -> Aggregate): a filter whose
 * `column` is the table's MvccData (HY_ENC_MVCC segments, see hy_validate) with predicate.condition = HY_FILTER_VALIDATE,
 * predicate.value.value_id = our transaction id, predicate.value2.value_id = the snapshot commit id and
 * predicate.column_is_nullable = can_use_chunk_shortcut keeps the rows hy_validate keeps (validate.cpp:47-68).
 * All columns are data columns (no reference segments) of one table.  HY_ERR_UNSUPPORTED: more than HY_MAX_FILTERS filters,
 * aggregate functions other than MIN / MAX / SUM / AVG / COUNT, string expressions, expressions that read more than six
 * distinct columns between them or need a fourth stack slot -- run the chain instead. */
enum { HY_EXPR_COLUMN = 0, HY_EXPR_LITERAL = 1, HY_EXPR_ARITHMETIC = 2 };
enum { HY_MAX_EXPRESSION_NODES = 12, HY_MAX_FILTERS = 4 };
enum { HY_FILTER_VALIDATE = 0x100 /* hy_filter.predicate.condition of a Validate filter; not a PredicateCondition */ };
typedef struct hy_expression_node {
  uint32_t kind;                  /* HY_EXPR_*                                                                         */
  uint32_t op;                    /* HY_EXPR_ARITHMETIC: HY_ARITH_*, applied to the two results below it on the stack  */
  const hy_column* column;        /* HY_EXPR_COLUMN                                                                    */
  uint32_t literal_type;          /* HY_EXPR_LITERAL: HY_TYPE_* (HY_TYPE_NULL: the NULL literal)                       */
  uint32_t reserved;
  hy_value literal;
} hy_expression_node;
typedef struct hy_expression {
  uint32_t n_nodes;
  uint32_t reserved;
  hy_expression_node nodes[HY_MAX_EXPRESSION_NODES];
} hy_expression;
typedef struct hy_filter {
  const hy_column* column;
  hy_predicate predicate;         /* as for hy_table_scan (literals already cast: hy_predicate_cast)                   */
} hy_filter;
typedef struct hy_fused_aggregate {
  uint32_t function;              /* HY_AGG_MIN / MAX / SUM / AVG / COUNT                                              */
  uint32_t reserved;
  const hy_expression* input;     /* NULL for COUNT(*)                                                                 */
} hy_fused_aggregate;
hy_status hy_scan_project_aggregate(const hy_filter* filters, uint32_t n_filters, const hy_column* const* groupby_columns, uint32_t n_groupby,
                                    const hy_fused_aggregate* aggregates, uint32_t n_aggregates, hy_aggregate_result* result);

/* ---- a star join as one call (BASELINE.json configs[4], the SSB star joins; csrc/plan.hip) ------------------------------------
 * The operator tree Hyrise runs for   fact JOIN dim_1 ... JOIN dim_k  GROUP BY ...  -- a TableScan per filtered dimension, one JoinHash
 * per dimension (the filtered dimension as build side, the join result so far as probe side, in the order given), a Projection for
 * the aggregates' arithmetic, an AggregateHash -- made by ONE function: exactly the calls of the entry points above an adapter would
 * make (hy_table_scan, hy_poslist_translate, hy_column_create, hy_column_export, hy_join_hash, hy_gather_row_ids,
 * hy_projection_arithmetic, hy_aggregate_hash), with every intermediate in device memory and no interpreter between them.  No
 * counterpart in the reference (its scheduler runs the operators one by one); the groups and their cells are those of the operator chain.
 * Where the shape allows (int32 primary keys with a range of at most 2^26 values, foreign keys as int32 values / FrameOfReference offsets
 * without NULLs) every dimension is probed in ONE pass over the fact table instead (csrc/join_star.hpp, HY_OPT_STAR_FUSED_PROBE): the join
 * result's rows -- and with them the groups, ordered by their first row -- then come in the fact table's row order, not in the order the
 * last JoinHash of the chain would leave them in.  Where, in addition, every GROUP BY column (one to four) and every aggregate input (at most
 * four aggregates: MIN / MAX / SUM / AVG / COUNT) is an int / long column, the Projection and the AggregateHash happen inside that join
 * (HY_OPT_STAR_FUSED_FINISH): the rows that survive every dimension are grouped where they are found; same groups, order, cells and
 * representative rows as the steps it replaces.
 * Columns are named as (table, column): table 0 = the fact table, d + 1 = dimension d; every column is a DATA column (numeric, or a
 * dictionary of key names / join ids) of its table.  An aggregate reads `left` alone (op = HY_STAR_NO_OP), `left <op> right`
 * (HY_ARITH_*: hy_projection_arithmetic), or nothing (left.column = NULL: COUNT(*)).  *joined_rows: rows of the join result.
 * NULLs: a foreign key that is NULL finds no partner, a NULL aggregate input is skipped, NULL GROUP BY cells are a group of their own -- as in
 * the operator chain (the plan's intermediate tables carry null vectors where a cell is NULL; such plans keep to the join-by-join / RowID paths). */
enum { HY_MAX_STAR_DIMENSIONS = 8, HY_MAX_STAR_AGGREGATES = 8 };
#define HY_STAR_NO_OP 0xFFFFFFFFu
typedef struct hy_star_dimension {
  const hy_column* key;            /* the dimension's key column: the build side of its join                                     */
  const hy_column* filter_column;  /* a column of the dimension that `predicate` tests; NULL: the dimension is not filtered        */
  hy_predicate predicate;
  const hy_column* fact_key;       /* the fact table's foreign key to this dimension                                             */
} hy_star_dimension;
typedef struct hy_star_column {
  uint32_t table;                  /* 0 = the fact table, d + 1 = dimension d                                                    */
  uint32_t reserved;
  const hy_column* column;
} hy_star_column;
typedef struct hy_star_aggregate {
  uint32_t function;               /* HY_AGG_*                                                                                   */
  uint32_t op;                     /* HY_STAR_NO_OP or HY_ARITH_*                                                                */
  hy_star_column left, right;
} hy_star_aggregate;
hy_status hy_star_join_aggregate(const hy_star_dimension* dimensions, uint32_t n_dimensions, const hy_star_column* groupby, uint32_t n_groupby,
                                 const hy_star_aggregate* aggregates, uint32_t n_aggregates, hy_aggregate_result* result, uint64_t* joined_rows);

/* ---- multi-GPU exchange (the reference is one process: these have no counterpart there) ----------------
 * Sharding an operator over GPUs adds one exchange step per operator (hyrise_amd/distributed.py: one process per GPU, RCCL):
 * the broadcast-build JoinHash all-gathers the build side's join column, the repartitioned JoinHash sends every (key, RowID)
 * to the GPU  std::hash(key) % parts  (the integer itself: JoinHash's own hash function, join_hash_steps.hpp:352), the
 * sharded AggregateHash all-reduces per-group partials.  The buffers of those collectives are produced and consumed on the
 * device by the entry points below; all pointers are DEVICE pointers unless stated. */
/* The column's values decoded into values[rows] (the column's type; NULL rows: 0) and, if nulls is not NULL, nulls[rows] bytes. */
hy_status hy_column_export(const hy_column* column, void* values, uint8_t* nulls);
/* Rows per destination of a hash repartition of an integer join column (NULL keys never find a partner: they are not sent).
 * counts[parts] is a HOST array. */
hy_status hy_repartition_count(const hy_column* column, uint32_t parts, uint64_t* counts);
/* The (key, RowID) tuples grouped by destination, (chunk, row) order inside each: keys_out (int32 for an int column, int64 for a
 * long column) and row_ids_out hold the tuples of destination 0, then 1, ...; counts[parts] (HOST) as above.  The RowIDs carry
 * chunk_id + chunk_id_offset (the shard's first chunk in the whole table). */
hy_status hy_repartition_pack(const hy_column* column, uint32_t parts, uint32_t chunk_id_offset, void* keys_out, hy_row_id* row_ids_out,
                              uint64_t capacity, uint64_t* counts);
/* out[i] = table[positions[i].chunk_id * chunk_rows + positions[i].chunk_offset] (NULL_ROW_ID stays): the PosList of a join over
 * received tuple arrays (presented as a column of chunk_rows-row chunks) back to the RowIDs that travelled with the keys. */
hy_status hy_gather_row_ids(const hy_row_id* table, uint64_t table_rows, uint32_t chunk_rows, const hy_row_id* positions, uint64_t n,
                            hy_row_id* out);

#ifdef __cplusplus
}
#endif
#endif /* HYRISE_AMD_H_ */
