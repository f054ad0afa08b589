// projection.hip -- the arithmetic a Projection evaluates.
//
// What it replaces (reference, CPU):
//   Projection::_on_execute -> ExpressionEvaluator::evaluate_expression_to_segment     operators/projection.cpp, expression/evaluation/expression_evaluator.cpp
//   _evaluate_arithmetic_expression + the functors of                                   expression/evaluation/expression_functors.hpp:127-213
//   expression_common_type                                                              expression/expression_utils.cpp:172-204
//
// The reference interprets the expression tree per chunk and materialises one ExpressionResult vector per node.  Here one
// kernel evaluates  left <op> right  for all chunks of a table (one workgroup per 8192-row slice, a lane per row, both
// operands decoded in place -- data or reference segments of any supported encoding, or a literal) and writes a
// device-resident column of unencoded value segments + null bitmap that the next operator consumes directly: for TPC-H Q6
// / Q1 the products never leave HBM.  HBM-bound: bytes in (operand widths) + bytes out (result width) per row.
#include "hy_device.hpp"
#include "hy_decode.hpp"
#include "hy_arithmetic.hpp"

#include <cstring>
#include <vector>

namespace hy {

struct Operand {
  const DevSegment* segments;   // nullptr: literal
  uint32_t type;                // HY_TYPE_*
  hy_value literal;
};

struct ProjectionArgs {
  Operand left, right;
  uint32_t op;
  uint32_t result_type;
  const Slice* slices;
  const uint64_t* row_base;     // first global row of every chunk
  void* values;                 // [rows] of the result type, chunk after chunk (chunk c at row_base[c], 16-byte padded per chunk: see value_base)
  const uint64_t* value_base;   // byte offset of every chunk's values
  uint64_t* nulls;              // bitmap words, chunk after chunk
  const uint64_t* null_base;    // word offset of every chunk's bitmap
};

// B rows of an operand as (bits, NULL mask): bits = int64 value or the bits of a double (float columns widened), like
// decode_rows; a literal is broadcast.
template <int B>
__device__ __forceinline__ void operand_rows(const Operand& o, uint32_t chunk, const uint32_t (&row)[B], uint32_t valid, uint64_t (&bits)[B], uint32_t* nulls) {
  if (o.segments) {
    // rows behind a single-chunk PosList (the output of a scan or a join) are read from the referenced segment at their offsets
    const uint32_t* pos_words;
    const DevSegment segment = resolve_segment(o.segments[chunk], &pos_words);
    uint32_t operand_row[B];
#pragma unroll
    for (int i = 0; i < B; ++i) operand_row[i] = row[i];
    const uint32_t null_rows = pos_words ? dereference_rows<B>(pos_words, operand_row) : 0u;
    decode_rows<B>(segment, o.segments, chunk, operand_row, valid, bits, nulls);
    *nulls |= null_rows;
    return;
  }
  uint64_t literal = 0;
  switch (o.type) {
    case HY_TYPE_INT: literal = static_cast<uint64_t>(static_cast<int64_t>(o.literal.i32)); break;
    case HY_TYPE_LONG: literal = static_cast<uint64_t>(o.literal.i64); break;
    case HY_TYPE_FLOAT: literal = static_cast<uint64_t>(__double_as_longlong(static_cast<double>(o.literal.f32))); break;
    case HY_TYPE_DOUBLE: literal = static_cast<uint64_t>(__double_as_longlong(o.literal.f64)); break;
    default: break;
  }
#pragma unroll
  for (int i = 0; i < B; ++i) bits[i] = literal;
  *nulls = o.type == HY_TYPE_NULL ? 0xFFFFFFFFu : 0u;
}

// ---- fast path: + - * over plain operands -----------------------------------------------------------------------------
// Both operands are literals or unencoded, aligned value segments without NULLs (what TPC-H's expressions over lineitem
// are), and the operator cannot produce a NULL: a thread takes four consecutive rows per step -- one 16-byte load per
// operand (two for 8-byte types), one or two 16-byte stores -- and the cell arithmetic is instantiated per (operator,
// operand types), so that no type dispatch is left inside the row loop.
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) pu32x4 global_pu32x4;

template <uint32_t TYPE>
__device__ __forceinline__ void load_four(const void* data, uint32_t row0, uint32_t n_valid, const hy_value& literal, bool is_literal, Value (&out)[4]) {
  if (!is_literal && n_valid < 4) {   // the group that straddles the end of the chunk: nothing is read behind the buffer
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[i] = Value{false, 0, 0.0};
      const uint32_t row = row0 + (static_cast<uint32_t>(i) < n_valid ? i : 0);
      if (TYPE == HY_TYPE_INT) out[i].i = static_cast<const int32_t*>(data)[row];
      else if (TYPE == HY_TYPE_LONG) out[i].i = static_cast<const int64_t*>(data)[row];
      else if (TYPE == HY_TYPE_FLOAT) out[i].f = static_cast<double>(static_cast<const float*>(data)[row]);
      else out[i].f = static_cast<const double*>(data)[row];
    }
    return;
  }
  if (is_literal) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[i] = Value{false, 0, 0.0};
      if (TYPE == HY_TYPE_INT) out[i].i = literal.i32;
      else if (TYPE == HY_TYPE_LONG) out[i].i = literal.i64;
      else if (TYPE == HY_TYPE_FLOAT) out[i].f = static_cast<double>(literal.f32);
      else out[i].f = literal.f64;
    }
    return;
  }
  if (TYPE == HY_TYPE_INT || TYPE == HY_TYPE_FLOAT) {
    const pu32x4 raw = reinterpret_cast<const global_pu32x4*>(reinterpret_cast<uintptr_t>(data))[row0 / 4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[i] = Value{false, 0, 0.0};
      if (TYPE == HY_TYPE_INT) out[i].i = static_cast<int32_t>(raw[i]);
      else out[i].f = static_cast<double>(__uint_as_float(raw[i]));
    }
  } else {
    const pu32x4 lo = reinterpret_cast<const global_pu32x4*>(reinterpret_cast<uintptr_t>(data))[row0 / 2];
    const pu32x4 hi = reinterpret_cast<const global_pu32x4*>(reinterpret_cast<uintptr_t>(data))[row0 / 2 + 1];
    const uint64_t bits[4] = {static_cast<uint64_t>(lo[1]) << 32 | lo[0], static_cast<uint64_t>(lo[3]) << 32 | lo[2], static_cast<uint64_t>(hi[1]) << 32 | hi[0],
                              static_cast<uint64_t>(hi[3]) << 32 | hi[2]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[i] = Value{false, 0, 0.0};
      if (TYPE == HY_TYPE_LONG) out[i].i = static_cast<int64_t>(bits[i]);
      else out[i].f = __longlong_as_double(static_cast<long long>(bits[i]));
    }
  }
}

__host__ __device__ constexpr uint32_t common_type_of(uint32_t lhs, uint32_t rhs) {   // expression_common_type for two numeric types
  if (lhs == HY_TYPE_DOUBLE || rhs == HY_TYPE_DOUBLE) return HY_TYPE_DOUBLE;
  if (lhs == HY_TYPE_LONG) return rhs == HY_TYPE_FLOAT ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
  if (rhs == HY_TYPE_LONG) return lhs == HY_TYPE_FLOAT ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
  if (lhs == HY_TYPE_FLOAT || rhs == HY_TYPE_FLOAT) return HY_TYPE_FLOAT;
  return HY_TYPE_INT;
}

// The stored words of four consecutive rows (row0 a multiple of four, all four inside the chunk): one 16-byte load, two for 8-byte types --
// unconditional, so that a thread's loads of several groups and of both operands are in flight together (a load under a condition makes the
// compiler wait for everything in flight where the paths meet: with one group per trip a wave had 16 bytes per lane in flight and the pass
// ran at 1.9 TB/s, profiles/r06_q1_chain_launches.txt).
struct RawFour { pu32x4 lo, hi; };
struct PlainNulls { const uint8_t *x, *y; };   // the operands' null bitmaps as bytes (nullptr: the operand has no NULL)
template <uint32_t TYPE>
__device__ __forceinline__ RawFour raw_four(const void* data, uint32_t row0) {
  RawFour r{};
  if (!data) return r;   // (a literal: uniform for the whole launch)
  if (TYPE == HY_TYPE_INT || TYPE == HY_TYPE_FLOAT) {
    r.lo = reinterpret_cast<const global_pu32x4*>(reinterpret_cast<uintptr_t>(data))[row0 / 4];
  } else {
    r.lo = reinterpret_cast<const global_pu32x4*>(reinterpret_cast<uintptr_t>(data))[row0 / 2];
    r.hi = reinterpret_cast<const global_pu32x4*>(reinterpret_cast<uintptr_t>(data))[row0 / 2 + 1];
  }
  return r;
}
template <uint32_t TYPE>
__device__ __forceinline__ void values_of(const RawFour& r, Value (&out)[4]) {
  const uint64_t wide[4] = {static_cast<uint64_t>(r.lo[1]) << 32 | r.lo[0], static_cast<uint64_t>(r.lo[3]) << 32 | r.lo[2], static_cast<uint64_t>(r.hi[1]) << 32 | r.hi[0],
                            static_cast<uint64_t>(r.hi[3]) << 32 | r.hi[2]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[i] = Value{false, 0, 0.0};
    if (TYPE == HY_TYPE_INT) out[i].i = static_cast<int32_t>(r.lo[i]);
    else if (TYPE == HY_TYPE_FLOAT) out[i].f = static_cast<double>(__uint_as_float(r.lo[i]));
    else if (TYPE == HY_TYPE_LONG) out[i].i = static_cast<int64_t>(wide[i]);
    else out[i].f = __longlong_as_double(static_cast<long long>(wide[i]));
  }
}

template <uint32_t OP, uint32_t AT, uint32_t BT>
__device__ __forceinline__ void plain_slice(const ProjectionArgs& a, const Slice& slice, const void* x_data, const void* y_data, char* values, const PlainNulls& operand_nulls) {
  constexpr uint32_t RT = common_type_of(AT, BT);
  constexpr bool WIDE = RT == HY_TYPE_LONG || RT == HY_TYPE_DOUBLE;
  constexpr uint32_t GROUPS = 4;   // groups of four rows a thread has in flight
#pragma unroll 1
  for (uint32_t base = 0; base < SLICE_ROWS / 1024; base += GROUPS) {
    RawFour x_raw[GROUPS], y_raw[GROUPS];
    uint32_t null_four[GROUPS];
#pragma unroll
    for (uint32_t g = 0; g < GROUPS; ++g) {   // the group that straddles the end of the chunk and the groups behind it ask the slice's first group (if that one is whole): nothing is read behind the buffer
      const uint32_t r0 = ((base + g) * 256 + threadIdx.x) * 4;
      const uint32_t from = slice.row_begin + (r0 + 3 < slice.row_count ? r0 : 0u);
      const bool readable = slice.row_count >= 4;
      x_raw[g] = raw_four<AT>(readable ? x_data : nullptr, from);
      y_raw[g] = raw_four<BT>(readable ? y_data : nullptr, from);
      typedef __attribute__((address_space(1))) const uint8_t global_byte;
      uint32_t null_byte = 0;   // the four rows' bits: one nibble of a byte of each bitmap (from is a multiple of four)
      if (operand_nulls.x) null_byte |= reinterpret_cast<global_byte*>(reinterpret_cast<uintptr_t>(operand_nulls.x))[from / 8];
      if (operand_nulls.y) null_byte |= reinterpret_cast<global_byte*>(reinterpret_cast<uintptr_t>(operand_nulls.y))[from / 8];
      null_four[g] = (null_byte >> (from & 4u)) & 0xFu;
    }
#pragma unroll
    for (uint32_t g = 0; g < GROUPS; ++g) {
      const uint32_t r0 = ((base + g) * 256 + threadIdx.x) * 4;
      if (r0 >= slice.row_count) continue;
      const uint32_t row0 = slice.row_begin + r0;
      Value x[4], y[4];
      const uint32_t n_valid = slice.row_count - r0 < 4 ? slice.row_count - r0 : 4;
      if (n_valid == 4) {
        if (x_data) values_of<AT>(x_raw[g], x);
        else load_four<AT>(x_data, row0, n_valid, a.left.literal, true, x);
        if (y_data) values_of<BT>(y_raw[g], y);
        else load_four<BT>(y_data, row0, n_valid, a.right.literal, true, y);
      } else {   // (one group per chunk at most: row by row)
        load_four<AT>(x_data, row0, n_valid, a.left.literal, x_data == nullptr, x);
        load_four<BT>(y_data, row0, n_valid, a.right.literal, y_data == nullptr, y);
        uint32_t null_byte = 0;
        if (operand_nulls.x) null_byte |= operand_nulls.x[row0 / 8];
        if (operand_nulls.y) null_byte |= operand_nulls.y[row0 / 8];
        null_four[g] = (null_byte >> (row0 & 4u)) & 0xFu;
      }
      uint64_t bits[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Value result{false, 0, 0.0};
        arithmetic_cell(OP, AT, BT, RT, x[i], y[i], &result);
        if (RT == HY_TYPE_INT) bits[i] = static_cast<uint32_t>(static_cast<int32_t>(result.i));
        else if (RT == HY_TYPE_LONG) bits[i] = static_cast<uint64_t>(result.i);
        else if (RT == HY_TYPE_FLOAT) bits[i] = __float_as_uint(static_cast<float>(result.f));
        else bits[i] = static_cast<uint64_t>(__double_as_longlong(result.f));
        if ((null_four[g] >> i) & 1u) bits[i] = 0;   // NULL cells hold T{} (value_segment.hpp)
      }
      const bool whole = r0 + 3 < slice.row_count;
      if (whole) {   // (the value buffers are 256-byte aligned per chunk, row0 is a multiple of four)
        if (WIDE) {
          pu32x4* out = reinterpret_cast<pu32x4*>(values) + row0 / 2;
          __builtin_nontemporal_store(pu32x4{static_cast<uint32_t>(bits[0]), static_cast<uint32_t>(bits[0] >> 32), static_cast<uint32_t>(bits[1]), static_cast<uint32_t>(bits[1] >> 32)}, out);
          __builtin_nontemporal_store(pu32x4{static_cast<uint32_t>(bits[2]), static_cast<uint32_t>(bits[2] >> 32), static_cast<uint32_t>(bits[3]), static_cast<uint32_t>(bits[3] >> 32)}, out + 1);
        } else {
          __builtin_nontemporal_store(pu32x4{static_cast<uint32_t>(bits[0]), static_cast<uint32_t>(bits[1]), static_cast<uint32_t>(bits[2]), static_cast<uint32_t>(bits[3])},
                                      reinterpret_cast<pu32x4*>(values) + row0 / 4);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (r0 + i >= slice.row_count) break;
          if (WIDE) reinterpret_cast<uint64_t*>(values)[row0 + i] = bits[i];
          else reinterpret_cast<uint32_t*>(values)[row0 + i] = static_cast<uint32_t>(bits[i]);
        }
      }
    }
  }
}

template <uint32_t OP, uint32_t AT>
__device__ __forceinline__ void plain_slice_by_right(const ProjectionArgs& a, const Slice& slice, const void* x, const void* y, char* values, const PlainNulls& n) {
  switch (a.right.type) {
    case HY_TYPE_INT: plain_slice<OP, AT, HY_TYPE_INT>(a, slice, x, y, values, n); break;
    case HY_TYPE_LONG: plain_slice<OP, AT, HY_TYPE_LONG>(a, slice, x, y, values, n); break;
    case HY_TYPE_FLOAT: plain_slice<OP, AT, HY_TYPE_FLOAT>(a, slice, x, y, values, n); break;
    default: plain_slice<OP, AT, HY_TYPE_DOUBLE>(a, slice, x, y, values, n); break;
  }
}

template <uint32_t OP>
__device__ __forceinline__ void plain_slice_by_types(const ProjectionArgs& a, const Slice& slice, const void* x, const void* y, char* values, const PlainNulls& n) {
  switch (a.left.type) {
    case HY_TYPE_INT: plain_slice_by_right<OP, HY_TYPE_INT>(a, slice, x, y, values, n); break;
    case HY_TYPE_LONG: plain_slice_by_right<OP, HY_TYPE_LONG>(a, slice, x, y, values, n); break;
    case HY_TYPE_FLOAT: plain_slice_by_right<OP, HY_TYPE_FLOAT>(a, slice, x, y, values, n); break;
    default: plain_slice_by_right<OP, HY_TYPE_DOUBLE>(a, slice, x, y, values, n); break;
  }
}

// operand of the fast path?  *data = its values in this chunk (nullptr: literal), *null_words = its bitmap (nullptr: none)
__device__ __forceinline__ bool plain_operand(const Operand& o, uint32_t chunk, const void** data, const uint64_t** null_words) {
  *data = nullptr;
  *null_words = nullptr;
  if (!o.segments) return o.type >= HY_TYPE_INT && o.type <= HY_TYPE_DOUBLE;   // (a NULL literal makes every cell NULL: generic path)
  const DevSegment& s = o.segments[chunk];
  if (s.encoding != HY_ENC_UNENCODED || (s.flags & SEG_UNALIGNED)) return false;
  *data = s.data;
  *null_words = s.nulls;   // (what an earlier projection leaves always carries a bitmap: Q1's charge = disc_price * (1 + l_tax) reads two of them)
  return true;
}

// One workgroup per 8192-row slice; a thread owns rows k*256 + tid (k = 0..31), decoded eight at a time (the loads of both
// operands first).  Row r of a wave's round lands in bit (r % 64) of one bitmap word: the null words are one ballot each.
__global__ __launch_bounds__(256) void projection_rows(ProjectionArgs a) {
  // Workgroup b runs on XCD b % 8, each with its own L2: of every 64 consecutive slices -- eight chunks of eight slices -- XCD x takes slices
  // 8x .. 8x + 7, one chunk, so that a chunk's dictionary is fetched into ONE L2 (aggregate_rows' mapping).  In slice order all eight L2s
  // gather from every chunk in flight: l_extendedprice's 240 KB dictionaries, 130 chunks at a time, do not fit, and the Q1 projection over
  // that column read 5.97 GB for 0.78 GB of operands (profiles/r06_q1_chain_launches.txt: 860 us against the other three's 350-400).
  uint32_t slice_index = blockIdx.x;
  if ((blockIdx.x | 63u) < gridDim.x) slice_index = (blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u);
  const Slice slice = a.slices[slice_index];
  const uint32_t lane = threadIdx.x & 63;
  char* values = static_cast<char*>(a.values) + a.value_base[slice.chunk];
  uint64_t* nulls = a.nulls + a.null_base[slice.chunk];
  const uint32_t at = a.left.type, bt = a.right.type, rt = a.result_type;
  if (a.op <= HY_ARITH_MUL) {
    const void *x_data, *y_data;
    const uint64_t *x_nulls, *y_nulls;
    if (plain_operand(a.left, slice.chunk, &x_data, &x_nulls) && plain_operand(a.right, slice.chunk, &y_data, &y_nulls)) {
      if (threadIdx.x < (slice.row_count + 63) / 64) {   // + - * make no NULL of their own: a cell is NULL where an operand's is (slices start at multiples of 8192: whole words)
        const uint32_t word = slice.row_begin / 64 + threadIdx.x;
        nulls[word] = (x_nulls ? x_nulls[word] : 0ull) | (y_nulls ? y_nulls[word] : 0ull);
      }
      const PlainNulls operand_nulls{reinterpret_cast<const uint8_t*>(x_nulls), reinterpret_cast<const uint8_t*>(y_nulls)};
      switch (a.op) {
        case HY_ARITH_ADD: plain_slice_by_types<HY_ARITH_ADD>(a, slice, x_data, y_data, values, operand_nulls); break;
        case HY_ARITH_SUB: plain_slice_by_types<HY_ARITH_SUB>(a, slice, x_data, y_data, values, operand_nulls); break;
        default: plain_slice_by_types<HY_ARITH_MUL>(a, slice, x_data, y_data, values, operand_nulls); break;
      }
      return;
    }
  }
  constexpr int B = 8;
#pragma unroll 1
  for (uint32_t block = 0; block < SLICE_ROWS / 256 / B; ++block) {
    uint32_t row[B], valid = 0;
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const uint32_t r = (block * B + i) * 256 + threadIdx.x;
      if (r < slice.row_count) valid |= 1u << i;
      row[i] = slice.row_begin + (r < slice.row_count ? r : 0);
    }
    uint64_t x[B], y[B];
    uint32_t x_nulls, y_nulls;
    operand_rows<B>(a.left, slice.chunk, row, valid, x, &x_nulls);
    operand_rows<B>(a.right, slice.chunk, row, valid, y, &y_nulls);
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const bool in = (valid >> i) & 1;
      bool is_null = ((x_nulls | y_nulls) >> i) & 1;
      Value result{false, 0, 0.0};
      if (in && !is_null) is_null = arithmetic_cell(a.op, at, bt, rt, as_value(x[i], at), as_value(y[i], bt), &result);
      if (in) {   // NULL cells hold T{} (value_segment.hpp)
        switch (rt) {
          case HY_TYPE_INT: reinterpret_cast<int32_t*>(values)[row[i]] = is_null ? 0 : static_cast<int32_t>(result.i); break;
          case HY_TYPE_LONG: reinterpret_cast<int64_t*>(values)[row[i]] = is_null ? 0 : result.i; break;
          case HY_TYPE_FLOAT: reinterpret_cast<float*>(values)[row[i]] = is_null ? 0.f : static_cast<float>(result.f); break;
          default: reinterpret_cast<double*>(values)[row[i]] = is_null ? 0.0 : result.f; break;
        }
      }
      const uint32_t r = (block * B + i) * 256 + threadIdx.x;
      const uint64_t null_lanes = __ballot(in && is_null);   // slices start at multiples of 8192: 64 consecutive rows = one word
      if (lane == 0 && r < ((slice.row_count + 63) & ~63u)) nulls[(slice.row_begin + r) >> 6] = null_lanes;
    }
  }
}

static bool numeric(uint32_t t) { return t >= HY_TYPE_INT && t <= HY_TYPE_DOUBLE; }

}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_projection_arithmetic(uint32_t op, const hy_operand* left, const hy_operand* right, hy_column** out) {
  if (!left || !right || !out) return fail(HY_ERR_INVALID, "hy_projection_arithmetic: null argument");
  *out = nullptr;
  if (op > HY_ARITH_MOD) return fail(HY_ERR_INVALID, "unknown arithmetic operator %u", op);
  HY_TRY(on_this_device(left->column, "hy_projection_arithmetic"));
  HY_TRY(on_this_device(right->column, "hy_projection_arithmetic"));
  hy_operand plain_left = *left, plain_right = *right;   // run-length / bit-packed segments: the decoded twins (hy_device.hpp)
  HY_TRY(plain_column(plain_left.column, &plain_left.column));
  HY_TRY(plain_column(plain_right.column, &plain_right.column));
  left = &plain_left;
  right = &plain_right;
  const hy_column* shape = left->column ? left->column : right->column;
  if (!shape) return fail(HY_ERR_INVALID, "hy_projection_arithmetic: at least one operand must be a column");
  for (const hy_operand* o : {left, right}) {
    if (o->column) {
      if (o->column->is_mvcc || (o->column->ref && o->column->ref->is_mvcc)) return fail(HY_ERR_INVALID, "MVCC columns are read by hy_validate only");
      if (!numeric(o->column->data_type)) return fail(HY_ERR_UNSUPPORTED, "string expressions stay on the CPU path");
      if (o->column->n_chunks != shape->n_chunks) return fail(HY_ERR_INVALID, "operand columns do not belong to one table (chunk counts differ)");
      for (uint32_t c = 0; c < shape->n_chunks; ++c) {
        if (o->column->host_segments[c].size != shape->host_segments[c].size) return fail(HY_ERR_INVALID, "operand columns do not belong to one table (chunk %u)", c);
      }
    } else if (o->literal_type != HY_TYPE_NULL && !numeric(o->literal_type)) {
      return fail(HY_ERR_UNSUPPORTED, "string expressions stay on the CPU path");
    }
  }
  const uint32_t left_type = left->column ? left->column->data_type : left->literal_type;
  const uint32_t right_type = right->column ? right->column->data_type : right->literal_type;
  if (left_type == HY_TYPE_NULL && right_type == HY_TYPE_NULL) return fail(HY_ERR_INVALID, "Cannot deduce common type if both sides are NULL.");
  const uint32_t result_type = expression_common_type(left_type, right_type);
  const uint32_t width = (result_type == HY_TYPE_INT || result_type == HY_TYPE_FLOAT) ? 4 : 8;
  hipStream_t stream = current_stream();

  // result buffers: one allocation, chunk c's values at value_base[c] (256-byte aligned), then all bitmaps
  const uint32_t n_chunks = shape->n_chunks;
  std::vector<uint64_t> value_base(n_chunks + 1, 0), null_base(n_chunks + 1, 0);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const uint64_t rows = shape->host_segments[c].size;
    value_base[c + 1] = value_base[c] + ((rows * width + 16 + 255) & ~uint64_t{255});
    null_base[c + 1] = null_base[c] + ((rows + 63) / 64 + 31) / 32 * 32;   // 256-byte aligned bitmaps
  }
  const uint64_t values_bytes = value_base[n_chunks], null_words = null_base[n_chunks];
  const uint64_t tables_bytes = 16 * (uint64_t{n_chunks} + 1);
  char* arena = nullptr;
  size_t arena_capacity = 0;
  HY_TRY(pool_acquire(values_bytes + 8 * null_words + tables_bytes + 256, reinterpret_cast<void**>(&arena), &arena_capacity));
  uint64_t* d_nulls = reinterpret_cast<uint64_t*>(arena + values_bytes);
  uint64_t* d_value_base = reinterpret_cast<uint64_t*>(arena + values_bytes + 8 * null_words);
  uint64_t* d_null_base = d_value_base + n_chunks + 1;
  auto release = [&](hy_status status) { pool_release(arena, arena_capacity); return status; };
  if (hipMemcpyAsync(d_value_base, value_base.data(), 8 * (size_t{n_chunks} + 1), hipMemcpyHostToDevice, stream) != hipSuccess ||
      hipMemcpyAsync(d_null_base, null_base.data(), 8 * (size_t{n_chunks} + 1), hipMemcpyHostToDevice, stream) != hipSuccess) {
    return release(fail(HY_ERR_DEVICE, "projection: upload of the chunk tables failed"));
  }
  ProjectionArgs a{};
  a.left = Operand{left->column ? left->column->d_segments : nullptr, left_type, left->literal};
  a.right = Operand{right->column ? right->column->d_segments : nullptr, right_type, right->literal};
  a.op = op;
  a.result_type = result_type;
  a.slices = shape->d_slices;
  a.row_base = shape->d_row_base;
  a.values = arena;
  a.value_base = d_value_base;
  a.nulls = d_nulls;
  a.null_base = d_null_base;
  if (shape->n_slices && shape->rows) {
    profile_begin(stream, HY_KERNEL_PROJECTION);
    hipLaunchKernelGGL(projection_rows, dim3(shape->n_slices), dim3(256), 0, stream, a);
    profile_end(stream);
  }
  if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return release(fail(HY_ERR_DEVICE, "projection kernel failed"));

  // the result as a column over the device buffers (HY_MEM_DEVICE: nothing is copied), which then owns them
  std::vector<hy_segment> segments(n_chunks ? n_chunks : 1);
  for (uint32_t c = 0; c < n_chunks; ++c) {
    hy_segment& s = segments[c];
    std::memset(&s, 0, sizeof(s));
    s.encoding = HY_ENC_UNENCODED;
    s.data_type = result_type;
    s.size = shape->host_segments[c].size;
    s.width = width;
    s.data = arena + value_base[c];
    s.nulls = d_nulls + null_base[c];
    s.ref_chunk_id = 0xFFFFFFFFu;
  }
  hy_column* column = nullptr;
  const hy_status status = hy_column_create(segments.data(), n_chunks, HY_MEM_DEVICE, &column);
  if (status != HY_OK) return release(status);
  column->pooled.emplace_back(arena_capacity, arena);
  *out = column;
  return HY_OK;
}

hy_status hy_column_read_chunk(const hy_column* column, uint32_t chunk, void* values, uint64_t* null_words) {
  if (!column || !values) return fail(HY_ERR_INVALID, "hy_column_read_chunk: null argument");
  if (chunk >= column->n_chunks) return fail(HY_ERR_INVALID, "chunk %u out of range", chunk);
  HY_TRY(plain_column(column, &column));
  const hy_segment& s = column->host_segments[chunk];
  if (s.encoding != HY_ENC_UNENCODED) return fail(HY_ERR_UNSUPPORTED, "hy_column_read_chunk reads unencoded value segments");
  hipStream_t stream = current_stream();
  if (s.size) HY_HIP(hipMemcpyAsync(values, s.data, size_t{s.width} * s.size, hipMemcpyDeviceToHost, stream));
  if (null_words) {
    const size_t words = (size_t{s.size} + 63) / 64;
    if (s.nulls && words) HY_HIP(hipMemcpyAsync(null_words, s.nulls, 8 * words, hipMemcpyDeviceToHost, stream));
    else std::memset(null_words, 0, 8 * words);
  }
  HY_HIP(hipStreamSynchronize(stream));
  return HY_OK;
}

uint32_t hy_column_data_type(const hy_column* column) { return column ? column->data_type : static_cast<uint32_t>(HY_TYPE_NULL); }

uint32_t hy_column_chunk_rows(const hy_column* column, uint32_t chunk) {
  return column && chunk < column->n_chunks ? column->host_segments[chunk].size : 0;
}

}  // extern "C"
