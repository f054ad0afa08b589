"""Multi-GPU execution: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  The reference is single-process; this is what SURVEY.md section 8(e) adds:

  TableScan      chunks shard naturally: rank r owns a contiguous chunk range, scans it alone, NO collective.
  AggregateHash  every rank aggregates its chunk range into partial (key, SUM, COUNT, MIN, MAX, first row) groups; ONE
                 all-gather of those few bytes (Q1: 4 groups x 6 aggregates) and a local, deterministic merge that
                 restores the reference's first-occurrence order (global first row = min over ranks).
  JoinHash       broadcast-build: the build side's join column is all-gathered (15 M keys x 4 B = 60 MB at SF10, arriving
                 over all 7 xGMI links at once), every rank builds the same table, the probe side stays chunk-sharded
                 and each rank emits the pairs of its probe chunks.  Pair order across ranks is the per-rank reference
                 order; the multi-GPU parity check is therefore on multisets (as the reference's own join tests are).

The per-rank work is done by an *executor* (the HIP library in production; the CPU oracle in the gloo tests, where no
GPU exists) -- this module only partitions, exchanges and merges.
"""
import numpy as np

from . import abi


def chunk_range(n_chunks, world_size, rank):
    """Contiguous chunk range [begin, end) of `rank`: ceil(C / G) chunks each."""
    per = (n_chunks + world_size - 1) // world_size if world_size else n_chunks
    begin = min(n_chunks, rank * per)
    return begin, min(n_chunks, begin + per)


def shard_column(host_column, world_size, rank):
    """The rank's chunks of a HostColumn (a view: segments are shared, not copied)."""
    from .storage import HostColumn
    begin, end = chunk_range(host_column.n_chunks, world_size, rank)
    return HostColumn(host_column.segments[begin:end], host_column.data_type), begin


def _all_gather_arrays(dist, array, device=None):
    """all_gather of a variable-length 1-D numpy array (lengths first, then padded payloads)."""
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    length = torch.tensor([array.size], dtype=torch.int64, device=dev)
    lengths = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lengths, length)
    lengths = [int(x.item()) for x in lengths]
    padded = np.zeros(max(lengths + [1]), dtype=array.dtype)
    padded[:array.size] = array
    local = torch.from_numpy(padded.view(np.uint8).copy()).to(dev)
    parts = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    return [p.cpu().numpy().view(array.dtype)[:n] for p, n in zip(parts, lengths)]


# ---- AggregateHash ---------------------------------------------------------------------------------------------------
def partial_aggregates(functions):
    """Per-rank aggregate list that carries enough to merge: AVG -> SUM(as double) + COUNT; everything else itself."""
    plan = []
    for f in functions:
        if f == abi.AGG_AVG:
            plan.append((abi.AGG_AVG, ("sum_as_double", "count")))
        else:
            plan.append((f, None))
    return plan


def merge_group_partials(parts, functions):
    """parts: per rank dict(keys=[tuple], first=[(global_chunk, offset)], values=[[per aggregate (value, count)]]).
    Returns merged groups in the reference's order (first occurrence over the whole table).  Deterministic: ranks are
    merged in rank order, so floating-point sums do not depend on arrival order."""
    merged = {}
    for part in parts:
        for key, first, values in zip(part["keys"], part["first"], part["values"]):
            entry = merged.get(key)
            if entry is None:
                merged[key] = {"first": first, "last": part.get("last", {}).get(key, first), "values": [list(v) for v in values]}
                continue
            entry["first"] = min(entry["first"], first)
            for a, (value, count) in enumerate(values):
                cur_value, cur_count = entry["values"][a]
                f = functions[a]
                if count == 0:
                    continue
                if cur_count == 0:
                    entry["values"][a] = [value, count]
                elif f == abi.AGG_MIN:
                    entry["values"][a] = [min(cur_value, value), cur_count + count]
                elif f == abi.AGG_MAX:
                    entry["values"][a] = [max(cur_value, value), cur_count + count]
                elif f in (abi.AGG_SUM, abi.AGG_AVG):
                    entry["values"][a] = [cur_value + value, cur_count + count]
                else:  # COUNT
                    entry["values"][a] = [0, cur_count + count]
    order = sorted(merged.items(), key=lambda kv: kv[1]["first"])
    return [(key, entry) for key, entry in order]


def sharded_aggregate(dist, executor, groupby_columns, aggregates, device=None):
    """groupby_columns / aggregates refer to the FULL table's HostColumns; every rank runs `executor` on its chunk range
    and the partials are exchanged with one all-gather.  executor(groupby, [(function, column)]) must return an object
    with n_groups, row_ids (representative = first row of the group in the shard), column(a) values and, for the
    merge, be called with SUM+COUNT for AVG (done here)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = {}

    def shard(col):
        if col is None:
            return None
        if id(col) not in shards:
            shards[id(col)] = shard_column(col, world, rank)
        return shards[id(col)][0]

    shape = groupby_columns[0] if groupby_columns else next(c for _, c in aggregates if c is not None)
    chunk_begin, _ = chunk_range(shape.n_chunks, world, rank)
    # local plan: every aggregate also needs its COUNT of non-NULL inputs; AVG is carried as SUM
    local = []
    for f, c in aggregates:
        local.append((abi.AGG_SUM if f == abi.AGG_AVG else f, shard(c)))
        local.append((abi.AGG_COUNT, shard(c)))
    gcols = [shard(c) for c in groupby_columns]
    rows_here = sum(s.size for s in (gcols[0].segments if gcols else next(c for _, c in local if c is not None).segments))
    keys, firsts, values = [], [], []
    if rows_here:
        # the group's key values are read back through ANY(group-by column)
        result = executor(gcols, local + [(abi.AGG_ANY, g) for g in gcols])
        n = result.n_groups
        cols = [result.column(i) for i in range(len(local) + len(gcols))]
        for g in range(n):
            key = tuple(cols[len(local) + k][g] for k in range(len(gcols)))
            keys.append(key)
            firsts.append((int(result.row_ids[g][0]) + chunk_begin, int(result.row_ids[g][1])))
            row = []
            for a, (f, _) in enumerate(aggregates):
                value, count = cols[2 * a][g], cols[2 * a + 1][g]
                if f == abi.AGG_AVG and value is not None:
                    value = float(value)
                row.append([0 if value is None else value, 0 if count is None else count])
            values.append(row)
    # exchange: one all-gather of a flat float64/int64 encoding (keys may be None -> flag)
    import pickle
    payload = np.frombuffer(pickle.dumps({"keys": keys, "first": firsts, "values": values}), dtype=np.uint8).copy()
    parts = [pickle.loads(p.tobytes()) for p in _all_gather_arrays(dist, payload, device)]
    functions = [f for f, _ in aggregates]
    merged = merge_group_partials(parts, functions)
    out_rows = []
    for key, entry in merged:
        row = list(key)
        for a, f in enumerate(functions):
            value, count = entry["values"][a]
            if f == abi.AGG_COUNT:
                row.append(count)
            elif count == 0:
                row.append(None)
            elif f == abi.AGG_AVG:
                row.append(value / count)
            else:
                row.append(value)
        out_rows.append((entry["first"], row))
    return out_rows


# ---- JoinHash --------------------------------------------------------------------------------------------------------
def gather_build_column(dist, build_values, build_nulls, device=None):
    """all-gather the build side's (decoded) join column: every rank ends up with the full build column, in rank order
    == chunk order."""
    values = _all_gather_arrays(dist, np.ascontiguousarray(build_values), device)
    nulls = None
    if build_nulls is not None:
        nulls = np.concatenate(_all_gather_arrays(dist, np.ascontiguousarray(build_nulls, dtype=np.uint8), device)).astype(bool)
    return np.concatenate(values), nulls
