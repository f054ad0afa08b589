// boundary.hip -- the host-side rules of the drop-in boundary that are plain arithmetic (no kernel): what the adapter does to
// a predicate's literals before a scan and to a join's PosLists before it builds output chunks.  They live behind the C ABI
// so that every host binding (the C++ operator mirror, the Python one, a cgo / JNI stub) applies the same rule.
//
//   hy_predicate_cast      TableScan::create_impl's literal handling (table_scan.cpp:336-366, 406-448) over
//                          lossless_predicate_variant_cast (utils/lossless_predicate_cast.cpp:40-73, .hpp:23-64) and
//                          lossless_cast (lossless_cast.hpp:32-187)
//   hy_join_output_chunks  write_output_chunks' partition merge (join_output_writing.cpp:245-296)
#include <cmath>
#include <cstring>
#include <limits>
#include <optional>
#include <type_traits>

#include "hy_device.hpp"

namespace hy {
namespace {

template <typename T> T read(const hy_value& v);
template <> int32_t read<int32_t>(const hy_value& v) { return v.i32; }
template <> int64_t read<int64_t>(const hy_value& v) { return v.i64; }
template <> float read<float>(const hy_value& v) { return v.f32; }
template <> double read<double>(const hy_value& v) { return v.f64; }

inline hy_value store(int32_t x) { hy_value v; std::memset(&v, 0, sizeof(v)); v.i32 = x; return v; }
inline hy_value store(int64_t x) { hy_value v; std::memset(&v, 0, sizeof(v)); v.i64 = x; return v; }
inline hy_value store(float x) { hy_value v; std::memset(&v, 0, sizeof(v)); v.f32 = x; return v; }
inline hy_value store(double x) { hy_value v; std::memset(&v, 0, sizeof(v)); v.f64 = x; return v; }

// The closest-to-zero values of the floating type that the integral type cannot hold (lossless_cast.hpp:137-158).
template <typename F, typename I> struct Bounds;
template <> struct Bounds<float, int32_t> { static constexpr float above = 2147483648.0f, below = -2147483904.0f; };
template <> struct Bounds<double, int32_t> { static constexpr double above = 2147483648.0, below = -2147483649.0; };
template <> struct Bounds<float, int64_t> { static constexpr float above = 9223372036854775808.0f, below = -9223373136366403584.0f; };
template <> struct Bounds<double, int64_t> { static constexpr double above = 9223372036854775808.0, below = -9223372036854777856.0; };

constexpr double LARGEST_FLOAT = 340282346638528859811704183484516925440.0;

// To `To` without losing information, or nothing.
template <typename To, typename From>
std::optional<To> exact(From x) {
  if constexpr (std::is_same_v<To, From>) {
    return x;
  } else if constexpr (std::is_integral_v<From> && std::is_integral_v<To>) {
    if constexpr (sizeof(From) > sizeof(To)) {
      if (x < static_cast<From>(std::numeric_limits<To>::min()) || x > static_cast<From>(std::numeric_limits<To>::max())) return std::nullopt;
    }
    return static_cast<To>(x);
  } else if constexpr (std::is_integral_v<From>) {          // integral -> floating: the round trip must give the integer back
    const To there = static_cast<To>(x);
    if (static_cast<From>(there) != x) return std::nullopt;
    return there;
  } else if constexpr (std::is_integral_v<To>) {            // floating -> integral: no fraction, inside the integral range
    From whole;
    if (std::modf(x, &whole) != From{0}) return std::nullopt;
    if (x >= Bounds<From, To>::above || x <= Bounds<From, To>::below) return std::nullopt;
    return static_cast<To>(x);
  } else if constexpr (sizeof(To) > sizeof(From)) {         // float -> double
    return static_cast<To>(x);
  } else {                                                  // double -> float
    if (x > LARGEST_FLOAT || x < -LARGEST_FLOAT) return std::nullopt;
    const float narrow = static_cast<float>(x);
    if (static_cast<double>(narrow) != x) return std::nullopt;
    return narrow;
  }
}

// The float next to the double `x` on the side of `towards` (next_float_towards, lossless_predicate_cast.cpp:14-38).
std::optional<float> neighbour_float(double x, double towards) {
  if (x > LARGEST_FLOAT || x < -LARGEST_FLOAT || x == towards) return std::nullopt;
  const float rounded = static_cast<float>(x);
  const double back = rounded;
  if ((back < x && towards < x) || (back > x && towards > x)) return rounded;   // rounding already went the right way
  const float next = std::nexttowardf(rounded, static_cast<long double>(towards));
  if (!std::isfinite(next)) return std::nullopt;
  return next;
}

struct Cast {
  bool ok = false;
  uint32_t condition = 0;
  hy_value value{};
};

template <typename To, typename From>
Cast cast_literal(uint32_t condition, From x) {
  Cast out;
  if (const auto same = exact<To>(x)) {
    out.ok = true, out.condition = condition, out.value = store(*same);
    return out;
  }
  if constexpr (std::is_same_v<From, double> && std::is_same_v<To, float>) {
    // x < 3.1 and x <= 3.1 hold for the same floats: those up to the largest float below 3.1 (and mirrored for > / >=);
    // = and != against a double that no float equals are left to the expression evaluator.
    const bool below = condition == HY_PRED_LESS_THAN || condition == HY_PRED_LESS_THAN_EQUALS;
    const bool above = condition == HY_PRED_GREATER_THAN || condition == HY_PRED_GREATER_THAN_EQUALS;
    if (below || above) {
      if (const auto bound = neighbour_float(x, below ? std::numeric_limits<double>::lowest() : std::numeric_limits<double>::max())) {
        out.ok = true, out.condition = below ? HY_PRED_LESS_THAN_EQUALS : HY_PRED_GREATER_THAN_EQUALS, out.value = store(*bound);
      }
    }
  }
  return out;
}

template <typename From>
Cast cast_to(uint32_t condition, From x, uint32_t target) {
  switch (target) {
    case HY_TYPE_INT: return cast_literal<int32_t>(condition, x);
    case HY_TYPE_LONG: return cast_literal<int64_t>(condition, x);
    case HY_TYPE_FLOAT: return cast_literal<float>(condition, x);
    case HY_TYPE_DOUBLE: return cast_literal<double>(condition, x);
    default: return Cast{};
  }
}

Cast cast_variant(uint32_t condition, uint32_t type, const hy_value& value, uint32_t target) {
  switch (type) {
    case HY_TYPE_INT: return cast_to(condition, read<int32_t>(value), target);
    case HY_TYPE_LONG: return cast_to(condition, read<int64_t>(value), target);
    case HY_TYPE_FLOAT: return cast_to(condition, read<float>(value), target);
    case HY_TYPE_DOUBLE: return cast_to(condition, read<double>(value), target);
    default: return Cast{};   // NULL literals and strings against numbers: no cast (lossless_predicate_cast.cpp:47-59)
  }
}

}  // namespace
}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_predicate_cast(uint32_t condition, uint32_t column_type, uint32_t literal_type, const hy_value* literal, uint32_t literal2_type,
                            const hy_value* literal2, hy_predicate* out) {
  if (!literal || !out) return fail(HY_ERR_INVALID, "hy_predicate_cast: null argument");
  std::memset(out, 0, sizeof(*out));
  const bool between = condition >= HY_PRED_BETWEEN_INCLUSIVE && condition <= HY_PRED_BETWEEN_EXCLUSIVE;
  if (!between && condition > HY_PRED_GREATER_THAN_EQUALS) return fail(HY_ERR_INVALID, "hy_predicate_cast: condition %u takes no numeric literal", condition);
  if (!between) {
    const Cast cast = cast_variant(condition, literal_type, *literal, column_type);
    if (!cast.ok) return fail(HY_ERR_UNSUPPORTED, "literal of type %u has no equivalent predicate on a column of type %u: the ExpressionEvaluator scan runs", literal_type, column_type);
    out->condition = cast.condition, out->value_type = column_type, out->value = cast.value;
    return HY_OK;
  }
  if (!literal2) return fail(HY_ERR_INVALID, "hy_predicate_cast: BETWEEN needs two literals");
  const bool lower_inclusive = condition == HY_PRED_BETWEEN_INCLUSIVE || condition == HY_PRED_BETWEEN_UPPER_EXCLUSIVE;
  const bool upper_inclusive = condition == HY_PRED_BETWEEN_INCLUSIVE || condition == HY_PRED_BETWEEN_LOWER_EXCLUSIVE;
  const Cast lower = cast_variant(lower_inclusive ? HY_PRED_GREATER_THAN_EQUALS : HY_PRED_GREATER_THAN, literal_type, *literal, column_type);
  const Cast upper = cast_variant(upper_inclusive ? HY_PRED_LESS_THAN_EQUALS : HY_PRED_LESS_THAN, literal2_type, *literal2, column_type);
  if (!lower.ok || !upper.ok) return fail(HY_ERR_UNSUPPORTED, "BETWEEN bounds of types %u / %u have no equivalent on a column of type %u: the ExpressionEvaluator scan runs", literal_type, literal2_type, column_type);
  const bool lower_closed = lower.condition == HY_PRED_GREATER_THAN_EQUALS, upper_closed = upper.condition == HY_PRED_LESS_THAN_EQUALS;
  out->condition = lower_closed ? (upper_closed ? HY_PRED_BETWEEN_INCLUSIVE : HY_PRED_BETWEEN_UPPER_EXCLUSIVE)
                                : (upper_closed ? HY_PRED_BETWEEN_LOWER_EXCLUSIVE : HY_PRED_BETWEEN_EXCLUSIVE);
  out->value_type = column_type, out->value = lower.value, out->value2 = upper.value;
  return HY_OK;
}

hy_status hy_join_output_chunks(const uint64_t* slice_offsets, uint32_t n_slices, uint64_t* chunk_offsets, uint32_t* n_chunks) {
  if (!slice_offsets || !chunk_offsets || !n_chunks) return fail(HY_ERR_INVALID, "hy_join_output_chunks: null argument");
  constexpr uint64_t MIN_SIZE = 1000, MAX_SIZE = 4 * MIN_SIZE;
  uint32_t chunks = 0;
  uint64_t begin = slice_offsets[0];       // first pair of the chunk being assembled
  chunk_offsets[0] = begin;
  bool open = false;                       // a chunk is being assembled (it may still take PosLists while below MIN_SIZE)
  for (uint32_t s = 0; s < n_slices; ++s) {
    const uint64_t size = slice_offsets[s + 1] - slice_offsets[s];
    const uint64_t have = slice_offsets[s] - begin;
    if (open && have < MIN_SIZE && have + size < MAX_SIZE) continue;   // PosList s joins the open chunk (empty ones, too)
    if (open) { chunk_offsets[++chunks] = slice_offsets[s]; open = false; }
    if (size == 0) continue;               // an empty PosList opens nothing
    begin = slice_offsets[s];
    open = true;
  }
  if (open) chunk_offsets[++chunks] = slice_offsets[n_slices];
  *n_chunks = chunks;
  return HY_OK;
}

}  // extern "C"
