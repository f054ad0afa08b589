/* hy_oracle: CPU restatement of the literal handling in front of a TableScan (TEST INFRASTRUCTURE, see hy_oracle.h).
 *
 * Follows  src/lib/lossless_cast.hpp:32-198            lossless_cast<Target>(Source) for the four numeric types
 *          src/lib/utils/lossless_predicate_cast.cpp:14-38   next_float_towards
 *          src/lib/utils/lossless_predicate_cast.hpp:23-64   lossless_predicate_cast<Output>(condition, input)
 *          src/lib/utils/lossless_predicate_cast.cpp:40-73   lossless_predicate_variant_cast
 *          src/lib/operators/table_scan.cpp:336-366,406-448  how TableScan::create_impl applies it to `column OP value`
 *                                                            and `column BETWEEN lower AND upper`
 *          src/lib/types.cpp:26-32,119-153                   is_binary_numeric_predicate_condition, between_to_conditions,
 *                                                            conditions_to_between
 * pinned by the reference's own known answers (lossless_predicate_cast_test.cpp, tests/golden).  String literals are outside
 * hy_value: a string literal against a string column is the identity, anything else involving strings is "no cast". */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "hy_oracle.h"

#define FLOAT_LIMIT 340282346638528859811704183484516925440.0   /* lossless_cast.hpp:178, lossless_predicate_cast.cpp:16 */

/* lossless_cast<target>(source value): 1 and *out on success, 0 for std::nullopt. */
static int lossless_cast(uint32_t source_type, const hy_value* in, uint32_t target_type, hy_value* out) {
  memset(out, 0, sizeof(*out));
  if (source_type == target_type) { *out = *in; return 1; }                                   /* identity, :32-36 */
  if (source_type == HY_TYPE_NULL || target_type == HY_TYPE_NULL) return 0;                  /* :55-66 */
  if (source_type == HY_TYPE_STRING || target_type == HY_TYPE_STRING) return 0;              /* (not representable here) */
  switch (source_type) {
    case HY_TYPE_LONG:
      if (target_type == HY_TYPE_INT) {                                                        /* :39-46 */
        if (in->i64 < INT32_MIN || in->i64 > INT32_MAX) return 0;
        out->i32 = (int32_t)in->i64;
        return 1;
      }
      if (target_type == HY_TYPE_FLOAT) { const float f = (float)in->i64; if ((int64_t)f != in->i64) return 0; out->f32 = f; return 1; }    /* :114-123 */
      { const double d = (double)in->i64; if ((int64_t)d != in->i64) return 0; out->f64 = d; return 1; }
    case HY_TYPE_INT:
      if (target_type == HY_TYPE_LONG) { out->i64 = in->i32; return 1; }                      /* :49-52 */
      if (target_type == HY_TYPE_FLOAT) { const float f = (float)in->i32; if ((int32_t)f != in->i32) return 0; out->f32 = f; return 1; }
      { const double d = (double)in->i32; if ((int32_t)d != in->i32) return 0; out->f64 = d; return 1; }
    case HY_TYPE_FLOAT:
      if (target_type == HY_TYPE_DOUBLE) { out->f64 = (double)in->f32; return 1; }            /* :164-168 */
      {                                                                                         /* :126-161 */
        float integral;
        if (modff(in->f32, &integral) != 0.0f) return 0;
        if (target_type == HY_TYPE_INT) { if (in->f32 >= 2147483648.0f || in->f32 <= -2147483904.0f) return 0; out->i32 = (int32_t)in->f32; return 1; }
        if (in->f32 >= 9223372036854775808.0f || in->f32 <= -9223373136366403584.0f) return 0;
        out->i64 = (int64_t)in->f32;
        return 1;
      }
    case HY_TYPE_DOUBLE:
      if (target_type == HY_TYPE_FLOAT) {                                                      /* :171-187 */
        if (in->f64 > FLOAT_LIMIT || in->f64 < -FLOAT_LIMIT) return 0;
        const float f = (float)in->f64;
        if ((double)f != in->f64) return 0;
        out->f32 = f;
        return 1;
      }
      {
        double integral;
        if (modf(in->f64, &integral) != 0.0) return 0;
        if (target_type == HY_TYPE_INT) { if (in->f64 >= 2147483648.0 || in->f64 <= -2147483649.0) return 0; out->i32 = (int32_t)in->f64; return 1; }
        if (in->f64 >= 9223372036854775808.0 || in->f64 <= -9223372036854777856.0) return 0;
        out->i64 = (int64_t)in->f64;
        return 1;
      }
    default:
      return 0;
  }
}

/* lossless_predicate_cast.cpp:14-38: the float next to the double `value` on the side of `towards` -- the rounded float itself when
 * rounding already went that way, else its neighbour in that direction; nothing for values outside the float range, for
 * value == towards, and when the neighbour is not finite. */
int hyo_next_float_towards(double value, double towards, float* out) {
  if (value > FLOAT_LIMIT || value < -FLOAT_LIMIT || value == towards) return 0;
  const float rounded = (float)value;
  const int downwards = towards < value;
  const int rounded_down = (double)rounded < value, rounded_up = (double)rounded > value;
  if ((downwards && rounded_down) || (!downwards && rounded_up)) {
    *out = rounded;
    return 1;
  }
  const float neighbour = nexttowardf(rounded, (long double)towards);
  if (!isfinite(neighbour)) return 0;
  *out = neighbour;
  return 1;
}

/* lossless_predicate_variant_cast (cpp:40-73) over lossless_predicate_cast<Output> (hpp:23-64). */
int hyo_lossless_predicate_cast(uint32_t condition, uint32_t source_type, const hy_value* value, uint32_t target_type, uint32_t* out_condition,
                                hy_value* out_value) {
  if (source_type == HY_TYPE_NULL || target_type == HY_TYPE_NULL) return 0;                  /* cpp:47-59 (NULL -> NULL is left to the evaluator, too) */
  if (lossless_cast(source_type, value, target_type, out_value)) {                            /* hpp:29-32 */
    *out_condition = condition;
    return 1;
  }
  if (condition > HY_PRED_GREATER_THAN_EQUALS) return 0;                                      /* hpp:34-36, types.cpp:26-32 */
  if (source_type == HY_TYPE_DOUBLE && target_type == HY_TYPE_FLOAT) {                        /* hpp:38-61 */
    float adjusted;
    if (condition == HY_PRED_EQUALS) return 0;
    if (condition == HY_PRED_LESS_THAN || condition == HY_PRED_LESS_THAN_EQUALS) {
      if (!hyo_next_float_towards(value->f64, -1.7976931348623157e308, &adjusted)) return 0;
      memset(out_value, 0, sizeof(*out_value));
      out_value->f32 = adjusted;
      *out_condition = HY_PRED_LESS_THAN_EQUALS;
      return 1;
    }
    if (condition == HY_PRED_GREATER_THAN || condition == HY_PRED_GREATER_THAN_EQUALS) {
      if (!hyo_next_float_towards(value->f64, 1.7976931348623157e308, &adjusted)) return 0;
      memset(out_value, 0, sizeof(*out_value));
      out_value->f32 = adjusted;
      *out_condition = HY_PRED_GREATER_THAN_EQUALS;
      return 1;
    }
  }
  return 0;                                                                                    /* hpp:63 (incl. double != float: NotEquals) */
}

/* TableScan::create_impl's literal handling: 1 = a ColumnVsValue / ColumnBetween scan runs with *out, 0 = the reference falls
 * back to the ExpressionEvaluator scan.  value2 == NULL for the binary conditions. */
int hyo_predicate_for_column(uint32_t condition, uint32_t column_type, uint32_t value_type, const hy_value* value, uint32_t value2_type,
                             const hy_value* value2, hy_predicate* out) {
  memset(out, 0, sizeof(*out));
  if (condition <= HY_PRED_GREATER_THAN_EQUALS) {                                              /* table_scan.cpp:356-366,381-384 */
    uint32_t adjusted;
    if (!hyo_lossless_predicate_cast(condition, value_type, value, column_type, &adjusted, &out->value)) return 0;
    out->condition = adjusted;
    out->value_type = column_type;
    return 1;
  }
  if (condition >= HY_PRED_BETWEEN_INCLUSIVE && condition <= HY_PRED_BETWEEN_EXCLUSIVE) {     /* :406-448 */
    /* between_to_conditions, types.cpp:119-132 */
    uint32_t lower = (condition == HY_PRED_BETWEEN_INCLUSIVE || condition == HY_PRED_BETWEEN_UPPER_EXCLUSIVE) ? HY_PRED_GREATER_THAN_EQUALS : HY_PRED_GREATER_THAN;
    uint32_t upper = (condition == HY_PRED_BETWEEN_INCLUSIVE || condition == HY_PRED_BETWEEN_LOWER_EXCLUSIVE) ? HY_PRED_LESS_THAN_EQUALS : HY_PRED_LESS_THAN;
    if (!value2) return 0;
    if (!hyo_lossless_predicate_cast(lower, value_type, value, column_type, &lower, &out->value)) return 0;
    if (!hyo_lossless_predicate_cast(upper, value2_type, value2, column_type, &upper, &out->value2)) return 0;
    /* conditions_to_between, types.cpp:134-153 (the casts only ever turn a strict bound into an inclusive one) */
    if (lower == HY_PRED_GREATER_THAN) out->condition = upper == HY_PRED_LESS_THAN ? HY_PRED_BETWEEN_EXCLUSIVE : HY_PRED_BETWEEN_LOWER_EXCLUSIVE;
    else out->condition = upper == HY_PRED_LESS_THAN ? HY_PRED_BETWEEN_UPPER_EXCLUSIVE : HY_PRED_BETWEEN_INCLUSIVE;
    out->value_type = column_type;
    return 1;
  }
  return 0;
}
