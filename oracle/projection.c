/* projection.c -- CPU restatement of the arithmetic that Hyrise's Projection evaluates through its ExpressionEvaluator
 * (test infrastructure, see hy_oracle.h).
 *
 *   result type        expression_common_type                         expression/expression_utils.cpp:172-204
 *   + - *              STLArithmeticFunctorWrapper: computed in std::common_type_t<A, B>, then cast to the result type
 *                                                                      expression/evaluation/expression_functors.hpp:127-150
 *   /                  DivisionEvaluator: NULL if an operand is NULL or the divisor is 0, else computed in the RESULT type
 *                                                                      expression_functors.hpp:188-213
 *   %                  ModuloEvaluator: NULL as above; `%` for two integral operands, std::fmod otherwise
 *                                                                      expression_functors.hpp:154-185
 * Pinned by the reference's known answers (expression_evaluator_to_values_test.cpp:228-256) in
 * tests/test_oracle_projection.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "hy_oracle.h"

uint32_t hyo_expression_common_type(uint32_t lhs, uint32_t rhs) {
  if (lhs == HY_TYPE_NULL) return rhs;
  if (rhs == HY_TYPE_NULL) return lhs;
  if (lhs == HY_TYPE_DOUBLE || rhs == HY_TYPE_DOUBLE) return HY_TYPE_DOUBLE;
  if (lhs == HY_TYPE_LONG) return (rhs == HY_TYPE_FLOAT) ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
  if (rhs == HY_TYPE_LONG) return (lhs == HY_TYPE_FLOAT) ? HY_TYPE_DOUBLE : HY_TYPE_LONG;
  if (lhs == HY_TYPE_FLOAT || rhs == HY_TYPE_FLOAT) return HY_TYPE_FLOAT;
  return HY_TYPE_INT;
}

/* std::common_type_t of the two C++ operand types (the usual arithmetic conversions) */
static uint32_t cxx_common_type(uint32_t a, uint32_t b) {
  if (a == HY_TYPE_DOUBLE || b == HY_TYPE_DOUBLE) return HY_TYPE_DOUBLE;
  if (a == HY_TYPE_FLOAT || b == HY_TYPE_FLOAT) return HY_TYPE_FLOAT;
  if (a == HY_TYPE_LONG || b == HY_TYPE_LONG) return HY_TYPE_LONG;
  return HY_TYPE_INT;
}

typedef struct { int64_t i; double f; } scalar_t; /* i for INT/LONG, f for FLOAT (rounded to float)/DOUBLE */

static int is_float_type(uint32_t t) { return t == HY_TYPE_FLOAT || t == HY_TYPE_DOUBLE; }

static scalar_t convert(scalar_t v, uint32_t from, uint32_t to) {
  scalar_t out = {0, 0.0};
  if (is_float_type(to)) {
    double d = is_float_type(from) ? v.f : (double)v.i;
    if (to == HY_TYPE_FLOAT) d = (double)(float)d;
    out.f = d;
  } else {
    int64_t i = is_float_type(from) ? (int64_t)v.f : v.i;
    if (to == HY_TYPE_INT) i = (int64_t)(int32_t)i;
    out.i = i;
  }
  return out;
}

static int is_zero(scalar_t v, uint32_t type) { return is_float_type(type) ? v.f == 0.0 : v.i == 0; }

/* one cell; returns 1 if the result is NULL */
int hyo_arithmetic_cell(uint32_t op, uint32_t a_type, const void* a, int a_null, uint32_t b_type, const void* b, int b_null, void* result) {
  const uint32_t result_type = hyo_expression_common_type(a_type, b_type);
  if (a_null || b_null || a_type == HY_TYPE_NULL || b_type == HY_TYPE_NULL) return 1;
  scalar_t x = {0, 0.0}, y = {0, 0.0};
  switch (a_type) { case HY_TYPE_INT: x.i = *(const int32_t*)a; break; case HY_TYPE_LONG: x.i = *(const int64_t*)a; break;
                    case HY_TYPE_FLOAT: x.f = *(const float*)a; break; default: x.f = *(const double*)a; break; }
  switch (b_type) { case HY_TYPE_INT: y.i = *(const int32_t*)b; break; case HY_TYPE_LONG: y.i = *(const int64_t*)b; break;
                    case HY_TYPE_FLOAT: y.f = *(const float*)b; break; default: y.f = *(const double*)b; break; }
  scalar_t r = {0, 0.0};
  uint32_t r_type = result_type;
  if (op == HY_ARITH_DIV) {
    if (is_zero(y, b_type)) return 1;
    const scalar_t p = convert(x, a_type, result_type), q = convert(y, b_type, result_type);
    if (is_float_type(result_type)) r.f = result_type == HY_TYPE_FLOAT ? (double)((float)p.f / (float)q.f) : p.f / q.f;
    else if (result_type == HY_TYPE_INT) r.i = (q.i == -1) ? (int64_t)(int32_t)(0u - (uint32_t)p.i) : (int64_t)((int32_t)p.i / (int32_t)q.i);
    else r.i = (q.i == -1) ? (int64_t)(0ull - (uint64_t)p.i) : p.i / q.i;
  } else if (op == HY_ARITH_MOD) {
    if (is_zero(y, b_type)) return 1;
    if (!is_float_type(a_type) && !is_float_type(b_type)) {
      const uint32_t c = cxx_common_type(a_type, b_type);
      r_type = c;
      r.i = (y.i == -1) ? 0 : (c == HY_TYPE_INT ? (int64_t)((int32_t)x.i % (int32_t)y.i) : x.i % y.i);
    } else if (a_type == HY_TYPE_FLOAT && b_type == HY_TYPE_FLOAT) {
      r_type = HY_TYPE_FLOAT;
      r.f = (double)fmodf((float)x.f, (float)y.f);
    } else { /* std::fmod with an integral or double argument: computed in double */
      r_type = HY_TYPE_DOUBLE;
      r.f = fmod(is_float_type(a_type) ? x.f : (double)x.i, is_float_type(b_type) ? y.f : (double)y.i);
    }
    r = convert(r, r_type, result_type);
  } else {
    const uint32_t c = cxx_common_type(a_type, b_type);
    const scalar_t p = convert(x, a_type, c), q = convert(y, b_type, c);
    if (c == HY_TYPE_DOUBLE) r.f = op == HY_ARITH_ADD ? p.f + q.f : op == HY_ARITH_SUB ? p.f - q.f : p.f * q.f;
    else if (c == HY_TYPE_FLOAT) {
      const float pf = (float)p.f, qf = (float)q.f;
      r.f = (double)(op == HY_ARITH_ADD ? pf + qf : op == HY_ARITH_SUB ? pf - qf : pf * qf);
    } else if (c == HY_TYPE_LONG) {
      const uint64_t pu = (uint64_t)p.i, qu = (uint64_t)q.i;
      r.i = (int64_t)(op == HY_ARITH_ADD ? pu + qu : op == HY_ARITH_SUB ? pu - qu : pu * qu);
    } else {
      const uint32_t pu = (uint32_t)p.i, qu = (uint32_t)q.i;
      r.i = (int64_t)(int32_t)(op == HY_ARITH_ADD ? pu + qu : op == HY_ARITH_SUB ? pu - qu : pu * qu);
    }
    r = convert(r, c, result_type);
  }
  switch (result_type) {
    case HY_TYPE_INT: *(int32_t*)result = (int32_t)r.i; break;
    case HY_TYPE_LONG: *(int64_t*)result = r.i; break;
    case HY_TYPE_FLOAT: *(float*)result = (float)r.f; break;
    default: *(double*)result = r.f; break;
  }
  return 0;
}

/* Element-wise over n cells; a literal operand has stride 0.  Returns the result type. */
uint32_t hyo_arithmetic(uint32_t op, uint32_t a_type, const void* a, const uint8_t* a_nulls, uint32_t a_stride, uint32_t b_type, const void* b,
                        const uint8_t* b_nulls, uint32_t b_stride, uint64_t n, void* result, uint8_t* result_nulls) {
  static const uint32_t width[6] = {0, 4, 8, 4, 8, 0};
  const uint32_t result_type = hyo_expression_common_type(a_type, b_type);
  for (uint64_t i = 0; i < n; ++i) {
    const int a_null = a_nulls ? a_nulls[a_stride ? i : 0] : 0, b_null = b_nulls ? b_nulls[b_stride ? i : 0] : 0;
    char* out = (char*)result + i * width[result_type];
    memset(out, 0, width[result_type]);
    result_nulls[i] = (uint8_t)hyo_arithmetic_cell(op, a_type, (const char*)a + (a_stride ? i : 0) * width[a_type], a_null, b_type,
                                                   (const char*)b + (b_stride ? i : 0) * width[b_type], b_null, out);
  }
  return result_type;
}
