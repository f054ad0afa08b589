// aggregate.hip -- AggregateHash on MI355X (placeholder until the kernels land).
#include "hy_device.hpp"
using namespace hy;
extern "C" {
hy_status hy_aggregate_hash(const hy_column* const*, uint32_t, const hy_aggregate_spec*, uint32_t, hy_aggregate_result*) { return fail(HY_ERR_UNSUPPORTED, "hy_aggregate_hash: not built yet"); }
}
