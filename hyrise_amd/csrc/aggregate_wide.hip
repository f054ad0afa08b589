// aggregate_wide.hip -- AggregateHash over 5 .. 8 GROUP BY columns: aggregate.hip compiled a second time with tuples of nine 64-bit words
// (NULL mask + eight values) instead of five.  The reference takes any number of GROUP BY columns (aggregate_hash.cpp:1184-1198: one to
// four columns get fixed-size keys, the `default:` case AggregateKeySmallVector; key construction :661-948); TPC-H Q10 groups by seven,
// Q18 by five.  The kernels are the same code; what changes is the register and LDS footprint of a tuple, which is why plans of up to
// four columns -- nearly all of them -- keep the build whose kernels are tuned for five words.  Entry points: hy_aggregate_hash_wide,
// hy_scan_project_aggregate_wide, reached through hy_aggregate_hash / hy_scan_project_aggregate when n_groupby > 4 (never called directly).
#define HY_MAX_GROUPBY 8
#include "aggregate.hip"
