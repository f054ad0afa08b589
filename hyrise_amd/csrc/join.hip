// join.hip -- JoinHash on MI355X (placeholder until the kernels land; every entry point reports UNSUPPORTED so the
// adapter keeps the stock CPU operator).
#include "hy_device.hpp"
using namespace hy;
extern "C" {
hy_status hy_join_hash(const hy_column*, const hy_column*, uint32_t, hy_join_result*) { return fail(HY_ERR_UNSUPPORTED, "hy_join_hash: not built yet"); }
hy_status hy_join_hash_count(const hy_column*, const hy_column*, uint32_t, uint64_t*) { return fail(HY_ERR_UNSUPPORTED, "hy_join_hash_count: not built yet"); }
hy_status hy_join_hash_radix_bits(uint64_t, uint64_t, uint32_t*) { return fail(HY_ERR_UNSUPPORTED, "not built yet"); }
}
