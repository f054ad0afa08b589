// hy_arithmetic.hpp -- one cell of the arithmetic a Projection evaluates: result types (expression_common_type,
// expression/expression_utils.cpp:172-204) and the functors of expression/evaluation/expression_functors.hpp:127-213.
// Shared by projection.hip (materialises the result column) and the fused scan -> projection -> aggregate kernel.
#pragma once
#include "hy_device.hpp"
#include "hy_decode.hpp"

namespace hy {

__device__ __forceinline__ bool is_float_type(uint32_t t) { return t == HY_TYPE_FLOAT || t == HY_TYPE_DOUBLE; }

// usual arithmetic conversions of the two C++ operand types (std::common_type_t)
__device__ __forceinline__ uint32_t cxx_common_type(uint32_t a, uint32_t b) {
  if (a == HY_TYPE_DOUBLE || b == HY_TYPE_DOUBLE) return HY_TYPE_DOUBLE;
  if (a == HY_TYPE_FLOAT || b == HY_TYPE_FLOAT) return HY_TYPE_FLOAT;
  if (a == HY_TYPE_LONG || b == HY_TYPE_LONG) return HY_TYPE_LONG;
  return HY_TYPE_INT;
}

__device__ __forceinline__ Value as_value(uint64_t bits, uint32_t type) {
  Value v{false, 0, 0.0};
  if (is_float_type(type)) v.f = __longlong_as_double(static_cast<long long>(bits));
  else v.i = static_cast<int64_t>(bits);
  return v;
}

// value of type `from` (Value convention: .i for integers, .f for float / double) as type `to`
__device__ __forceinline__ Value convert(const Value& v, uint32_t from, uint32_t to) {
  Value out{false, 0, 0.0};
  if (is_float_type(to)) {
    double d = is_float_type(from) ? v.f : static_cast<double>(v.i);
    if (to == HY_TYPE_FLOAT) d = static_cast<double>(static_cast<float>(d));
    out.f = d;
  } else {
    int64_t i = is_float_type(from) ? static_cast<int64_t>(v.f) : v.i;
    if (to == HY_TYPE_INT) i = static_cast<int64_t>(static_cast<int32_t>(i));
    out.i = i;
  }
  return out;
}

// one cell: returns true if the result is NULL (division / modulo by zero)
__device__ __forceinline__ bool arithmetic_cell(uint32_t op, uint32_t at, uint32_t bt, uint32_t rt, const Value& x, const Value& y, Value* out) {
  Value result{false, 0, 0.0};
  if (op == HY_ARITH_DIV || op == HY_ARITH_MOD) {
    if (is_float_type(bt) ? y.f == 0.0 : y.i == 0) return true;   // expression_functors.hpp:174,205
  }
  if (op == HY_ARITH_DIV) {          // computed in the result type
    const Value p = convert(x, at, rt), q = convert(y, bt, rt);
    if (rt == HY_TYPE_DOUBLE) result.f = p.f / q.f;
    else if (rt == HY_TYPE_FLOAT) result.f = static_cast<double>(static_cast<float>(p.f) / static_cast<float>(q.f));
    else if (rt == HY_TYPE_INT) result.i = q.i == -1 ? static_cast<int64_t>(static_cast<int32_t>(0u - static_cast<uint32_t>(p.i)))
                                                     : static_cast<int64_t>(static_cast<int32_t>(p.i) / static_cast<int32_t>(q.i));
    else result.i = q.i == -1 ? static_cast<int64_t>(0ull - static_cast<uint64_t>(p.i)) : p.i / q.i;
  } else if (op == HY_ARITH_MOD) {
    uint32_t computed = rt;
    if (!is_float_type(at) && !is_float_type(bt)) {
      computed = cxx_common_type(at, bt);
      result.i = y.i == -1 ? 0 : (computed == HY_TYPE_INT ? static_cast<int64_t>(static_cast<int32_t>(x.i) % static_cast<int32_t>(y.i)) : x.i % y.i);
    } else if (at == HY_TYPE_FLOAT && bt == HY_TYPE_FLOAT) {
      computed = HY_TYPE_FLOAT;
      result.f = static_cast<double>(fmodf(static_cast<float>(x.f), static_cast<float>(y.f)));
    } else {   // std::fmod with an integral or double argument: in double
      computed = HY_TYPE_DOUBLE;
      result.f = fmod(is_float_type(at) ? x.f : static_cast<double>(x.i), is_float_type(bt) ? y.f : static_cast<double>(y.i));
    }
    result = convert(result, computed, rt);
  } else {                              // + - * : computed in the common C++ type, cast to the result type
    const uint32_t c = cxx_common_type(at, bt);
    const Value p = convert(x, at, c), q = convert(y, bt, c);
    if (c == HY_TYPE_DOUBLE) {
      result.f = op == HY_ARITH_ADD ? __dadd_rn(p.f, q.f) : op == HY_ARITH_SUB ? __dsub_rn(p.f, q.f) : __dmul_rn(p.f, q.f);
    } else if (c == HY_TYPE_FLOAT) {
      const float pf = static_cast<float>(p.f), qf = static_cast<float>(q.f);
      // single IEEE operations: the compiler must not contract them with neighbouring operations
      const float rf = op == HY_ARITH_ADD ? __fadd_rn(pf, qf) : op == HY_ARITH_SUB ? __fsub_rn(pf, qf) : __fmul_rn(pf, qf);
      result.f = static_cast<double>(rf);
    } else if (c == HY_TYPE_LONG) {
      const uint64_t pu = static_cast<uint64_t>(p.i), qu = static_cast<uint64_t>(q.i);
      result.i = static_cast<int64_t>(op == HY_ARITH_ADD ? pu + qu : op == HY_ARITH_SUB ? pu - qu : pu * qu);
    } else {
      const uint32_t pu = static_cast<uint32_t>(p.i), qu = static_cast<uint32_t>(q.i);
      result.i = static_cast<int64_t>(static_cast<int32_t>(op == HY_ARITH_ADD ? pu + qu : op == HY_ARITH_SUB ? pu - qu : pu * qu));
    }
    result = convert(result, c, rt);
  }
  *out = result;
  return false;
}

inline uint32_t expression_common_type(uint32_t lhs, uint32_t rhs) {   // expression_utils.cpp:172-204
  if (lhs == HY_TYPE_NULL) return rhs;
  if (rhs == HY_TYPE_NULL) return lhs;
  if (lhs == HY_TYPE_DOUBLE || rhs == HY_TYPE_DOUBLE) return HY_TYPE_DOUBLE;
  if (lhs == HY_TYPE_LONG) return rhs == HY_TYPE_FLOAT ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
  if (rhs == HY_TYPE_LONG) return lhs == HY_TYPE_FLOAT ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
  if (lhs == HY_TYPE_FLOAT || rhs == HY_TYPE_FLOAT) return HY_TYPE_FLOAT;
  return HY_TYPE_INT;
}

}  // namespace hy
