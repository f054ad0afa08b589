// aggregate_widest.hip -- AggregateHash over 9 .. 16 GROUP BY columns: aggregate.hip compiled a third time, with tuples of seventeen 64-bit
// words (NULL mask + sixteen values).  The reference takes any number of GROUP BY columns (aggregate_hash.cpp:1184-1198: the `default:` case,
// AggregateKeySmallVector; key construction :661-948).  Plans this wide are rare (a SELECT DISTINCT over a wide projection, a GROUP BY that
// drags a table's descriptive columns along, like TPC-H Q10 with a few more): they run on aggregate_rows / fused_rows and the global group
// table only -- no partitioned path -- which is correct for any number of groups and fast for up to a few thousand.  Entry points:
// hy_aggregate_hash_widest, hy_scan_project_aggregate_widest, reached through the nine-word build when n_groupby > 8 (never called directly).
#define HY_MAX_GROUPBY 16
#include "aggregate.hip"
