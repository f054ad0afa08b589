"""Multi-GPU execution: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the
CPU tests and in the two-ranks-on-one-GPU debug mode).  The reference is single-process; this is what SURVEY.md section 8(e)
adds -- one exchange step per operator, on device buffers:

  TableScan      chunks shard naturally: rank r owns a contiguous chunk range, scans it alone, NO collective.
  AggregateHash  every rank aggregates its chunk range (hy_aggregate_hash) into per-group partials -- SUM and COUNT (AVG is
                 their quotient), MIN, MAX, the group's first row.  Small canonical key domains (TPC-H Q1: 4 groups) go into
                 FIXED SLOTS, slot = mixed-radix index of the key tuple in the all-reduced key ranges, and the partial arrays
                 are combined by all-reduce (SUM / MIN / MAX): a few hundred bytes, latency-bound.  Otherwise the groups'
                 (key, partial) arrays are all-gathered and merged locally.  The reference's group order is restored from the
                 all-reduced first rows (first occurrence) or keys (immediate-key shortcut, aggregate_hash.cpp:770-804).
  JoinHash (A)   broadcast-build: the build side's join column is decoded on the device (hy_column_export), all-gathered
                 (15 M keys x 4 B = 60 MB at SF10, arriving over all 7 xGMI links at once), every rank joins the whole build
                 side with its probe chunks.
  JoinHash (B)   hash repartition: both sides send every (key, RowID) to GPU  key % G  (hy_repartition_pack -> all-to-all),
                 every GPU joins what it received and maps the positions back to the RowIDs that travelled
                 (hy_gather_row_ids).  For build sides that do not fit one GPU; BASELINE.json names it for the SSB star join.
  Pair order across ranks is per-rank reference order (ranks in order); the N > 1 parity checks are on multisets, as the
  reference's own join tests are (join_test_runner.cpp:786).

The per-rank work is done by an *executor*: `HipExecutor` (the C ABI on this rank's GPU) in production and in the GPU tests,
an oracle-backed one in the CPU tests (tests/test_distributed_cpu.py), where no GPU exists.  This module only partitions,
exchanges and merges; collectives move torch tensors that live where the executor's buffers live.
"""
import ctypes as C

import numpy as np

from . import abi

FIXED_SLOT_LIMIT = 4096      # key domains up to this many slots are combined by all-reduce
REPARTITION_CHUNK = 65535    # received tuple arrays are presented as columns of this chunk size (Chunk::DEFAULT_SIZE)
_NULL_ROW = -1               # 0xFFFFFFFF as int32


def chunk_range(n_chunks, world_size, rank):
    """Contiguous chunk range [begin, end) of `rank`: ceil(C / G) chunks each."""
    per = (n_chunks + world_size - 1) // world_size if world_size else n_chunks
    begin = min(n_chunks, rank * per)
    return begin, min(n_chunks, begin + per)


def shard_column(host_column, world_size, rank):
    """The rank's chunks of a HostColumn (a view: segments are shared, not copied) and its first chunk id."""
    from .storage import HostColumn
    begin, end = chunk_range(host_column.n_chunks, world_size, rank)
    return HostColumn(host_column.segments[begin:end], host_column.data_type), begin


# ---- collectives -----------------------------------------------------------------------------------------------------
class Comm:
    """torch.distributed with the two things the operators need on top: variable-length all-gather / all-to-all, and -- for
    the gloo backend over CUDA tensors (several ranks sharing one GPU in the debug mode) -- staging through host memory."""

    def __init__(self, dist):
        import torch
        self.dist, self.torch = dist, torch
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.staged = dist.get_backend() == "gloo"

    def _in(self, t):
        return t.cpu() if self.staged and t.is_cuda else t

    def all_reduce(self, t, op):
        """In place; op: "sum" | "min" | "max"."""
        reduce_op = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}[op]
        staged = self._in(t)
        self.dist.all_reduce(staged, op=reduce_op)
        if staged is not t:
            t.copy_(staged)
        return t

    def all_gather_counts(self, n):
        torch = self.torch
        mine = torch.tensor([int(n)], dtype=torch.int64)
        if not self.staged:
            mine = mine.to(self._device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [int(x.item()) for x in out]

    def bind(self, device):
        self._device = device
        return self

    def all_gather_var(self, t):
        """Every rank's 1-D (or [n, k]) tensor, in rank order."""
        torch = self.torch
        counts = self.all_gather_counts(t.shape[0])
        width = max(counts + [1])
        staged = self._in(t)
        padded = torch.zeros((width,) + tuple(t.shape[1:]), dtype=t.dtype, device=staged.device)
        padded[:t.shape[0]] = staged
        parts = [torch.zeros_like(padded) for _ in range(self.world)]
        self.dist.all_gather(parts, padded)
        return [p[:n].to(t.device) for p, n in zip(parts, counts)]

    def all_to_all_var(self, send, send_counts):
        """send: rows grouped by destination (send_counts[d] rows for rank d).  Returns (received rows grouped by source,
        counts per source)."""
        torch = self.torch
        counts = torch.tensor([int(c) for c in send_counts], dtype=torch.int64)
        if not self.staged:
            counts = counts.to(self._device)
        received = torch.zeros_like(counts)
        self.dist.all_to_all_single(received, counts)
        recv_counts = [int(x) for x in received.cpu().tolist()]
        staged = self._in(send).contiguous()
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=staged.device)
        self.dist.all_to_all_single(out, staged, output_split_sizes=recv_counts, input_split_sizes=[int(c) for c in send_counts])
        return out.to(send.device), recv_counts


# ---- the per-rank executor over the C ABI ---------------------------------------------------------------------------------
class DeviceValueColumn:
    """An hy_column over a torch tensor of values in device memory (HY_MEM_DEVICE: nothing is copied), chunked like a table
    with `chunk_rows` rows per chunk; keeps the tensor(s) alive."""

    def __init__(self, lib, values, chunk_rows, data_type, null_bytes=None):
        import torch
        self.lib, self.values, self.data_type = lib, values, data_type
        self.rows = int(values.shape[0])
        width = values.element_size()
        self.null_words = None
        if null_bytes is not None and bool(null_bytes.any().item()):
            bits = null_bytes.to(torch.uint8)
            padded = torch.zeros(((self.rows + 63) // 64) * 64, dtype=torch.uint8, device=bits.device)
            padded[:self.rows] = bits
        else:
            padded = None
        n_chunks = max(1, (self.rows + chunk_rows - 1) // chunk_rows)   # (no rows: ONE empty chunk, so that the column still has its data type)
        self.n_chunks = n_chunks
        segments = (abi.Segment * max(1, n_chunks))()
        self._null_chunks = []
        for c in range(n_chunks):
            begin, end = c * chunk_rows, min(self.rows, (c + 1) * chunk_rows)
            s = segments[c]
            s.encoding, s.data_type, s.size, s.width = abi.ENC_UNENCODED, data_type, end - begin, width
            s.data = values.data_ptr() + begin * width
            s.ref_chunk_id = abi.INVALID_CHUNK_ID
            if padded is not None:   # the chunk's bits as libstdc++ vector<bool> words (little endian bytes of LSB-first bits)
                chunk_bits = torch.zeros(((end - begin + 63) // 64) * 64, dtype=torch.uint8, device=bits.device)
                chunk_bits[:end - begin] = bits[begin:end]
                weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=bits.device)
                packed = (chunk_bits.view(-1, 8).to(torch.int32) * weights).sum(dim=1).to(torch.uint8).contiguous()
                self._null_chunks.append(packed)
                s.nulls = packed.data_ptr()
        self._segments = segments
        handle = C.c_void_p()
        abi.check(lib.hy_column_create(segments, n_chunks, abi.MEM_DEVICE, C.byref(handle)))
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hy_column_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceReferenceColumn:
    """A column of a reference table in device memory: `rows` ([n, 2] int32 RowIDs into `base`, a data column) presented as
    ReferenceSegments of `chunk_rows` rows each (HY_MEM_DEVICE: the PosList a join or scan just wrote is read in place)."""

    def __init__(self, lib, base, rows, chunk_rows):
        import torch
        self.lib, self.base, self.data_type = lib, base, base.data_type
        if rows.shape[0] == 0:   # an empty table still has a type: one empty chunk
            rows = torch.zeros((1, 2), dtype=torch.int32, device=rows.device)
            self.rows = 0
        else:
            rows = rows.contiguous()
            self.rows = int(rows.shape[0])
        self.pos = rows
        n_chunks = max(1, (self.rows + chunk_rows - 1) // chunk_rows)
        self.n_chunks = n_chunks
        segments = (abi.Segment * n_chunks)()
        for c in range(n_chunks):
            begin, end = c * chunk_rows, min(self.rows, (c + 1) * chunk_rows)
            s = segments[c]
            s.encoding, s.data_type, s.size, s.width = abi.ENC_REFERENCE, base.data_type, max(0, end - begin), 8
            s.data = rows.data_ptr() + begin * 8
            s.ref = base.handle
            s.ref_chunk_id = abi.INVALID_CHUNK_ID
        self._segments = segments
        handle = C.c_void_p()
        abi.check(lib.hy_column_create(segments, n_chunks, abi.MEM_DEVICE, C.byref(handle)))
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hy_column_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SEGMENT_DTYPE = np.dtype([("encoding", np.uint32), ("data_type", np.uint32), ("size", np.uint32), ("width", np.uint32), ("data", np.uint64), ("aux", np.uint64),
                           ("aux_size", np.uint32), ("ref_chunk_id", np.uint32), ("nulls", np.uint64), ("ref", np.uint64), ("sorted_by", np.uint32), ("bits", np.uint32)])
assert _SEGMENT_DTYPE.itemsize == C.sizeof(abi.Segment)


class DevicePosLists:
    """A scan's output in the reference's shape: one PosList per input chunk, in device memory (`rows`: [input rows, 2] int32,
    chunk c's PosList at rows[begin[c] : begin[c] + count[c]], RowIDs of the DATA table `base_chunk[c]` -- the one chunk it
    references -- or mixed when base_chunk[c] is INVALID_CHUNK_ID).  The counts are the only thing the host knows of it."""

    def __init__(self, rows, begin, count, base_chunk):
        self.rows, self.begin, self.count, self.base_chunk = rows, begin, count, base_chunk
        self.total = int(count.sum())


class DeviceChunkedReferenceColumn:
    """Column `base` seen through DevicePosLists: one ReferenceSegment per non-empty PosList, read in place (HY_MEM_DEVICE), each
    with the single-chunk guarantee the PosLists carry (ReferenceSegment + RowIDPosList::guarantee_single_chunk)."""

    def __init__(self, lib, base, pos_lists):
        self.lib, self.base, self.data_type, self.pos = lib, base, base.data_type, pos_lists
        keep = np.flatnonzero(pos_lists.count)
        self.chunk_of_segment = keep                      # output chunk k <- input chunk keep[k]
        self.rows = pos_lists.total
        self.n_chunks = max(1, len(keep))
        # the hy_segment array, filled column-wise (one ctypes store per field and chunk costs more than the scan itself)
        table = np.zeros(self.n_chunks, dtype=_SEGMENT_DTYPE)
        table["encoding"], table["data_type"], table["width"] = abi.ENC_REFERENCE, base.data_type, 8
        table["ref"] = base.handle.value if isinstance(base.handle, C.c_void_p) else int(base.handle)
        if len(keep) == 0:   # an empty table still has a type: one empty chunk
            self._empty = pos_lists.rows.new_zeros((1, 2))
            table["data"], table["ref_chunk_id"] = self._empty.data_ptr(), abi.INVALID_CHUNK_ID
        else:
            table["size"] = pos_lists.count[keep]
            table["data"] = pos_lists.rows.data_ptr() + pos_lists.begin[keep].astype(np.uint64) * 8
            table["ref_chunk_id"] = pos_lists.base_chunk[keep]
        segments = table.ctypes.data_as(C.POINTER(abi.Segment))
        self._table = table
        self._segments = segments
        handle = C.c_void_p()
        abi.check(lib.hy_column_create(segments, self.n_chunks, abi.MEM_DEVICE, C.byref(handle)))
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hy_column_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _is_data_column(column):
    """A DeviceColumn over data segments (not a reference table, not MvccData): its scan result needs no dereferencing."""
    host = getattr(column, "host", None)
    return host is not None and all(s.encoding not in (abi.ENC_REFERENCE, abi.ENC_MVCC) for s in host.segments)


class HipExecutor:
    """Everything a rank computes, through libhyrise_amd.so on its GPU.  Columns are DeviceColumn / DeviceValueColumn /
    DeviceReferenceColumn / operators.ResultColumn."""
    _TORCH = None

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device
        self.lib = abi.load_library()
        self._types = {abi.TYPE_INT: torch.int32, abi.TYPE_LONG: torch.int64, abi.TYPE_FLOAT: torch.float32, abi.TYPE_DOUBLE: torch.float64}

    def column(self, host_column):
        from .storage import DeviceColumn
        return DeviceColumn(host_column)

    def rows_of(self, column):
        return column.rows

    def aggregate(self, groupby, aggregates):
        from .operators import aggregate_hash
        shape = groupby[0] if groupby else next(c for _, c in aggregates if c is not None)
        try:   # few groups are the rule; a result that does not fit says so
            return aggregate_hash(groupby, aggregates, group_capacity=min(shape.rows + 1, 4096))
        except abi.HyriseAmdError as error:
            if error.status != abi.ERR_CAPACITY:
                raise
        return aggregate_hash(groupby, aggregates, group_capacity=shape.rows + 1)

    def scan_project_aggregate(self, filters, groupby, aggregates):
        """TableScan(s) -> Projection -> AggregateHash of this rank's chunks in one pass (hy_scan_project_aggregate); aggregates:
        [(function, expression tree or None)], trees as operators.expression takes them."""
        from .operators import scan_project_aggregate
        try:
            return scan_project_aggregate(filters, groupby, aggregates, group_capacity=4096)
        except abi.HyriseAmdError as error:
            if error.status != abi.ERR_CAPACITY:
                raise
        return scan_project_aggregate(filters, groupby, aggregates)

    def _device_scan(self, column, predicate, layout, visibility=None):
        """hy_table_scan (or, with visibility = (our_tid, snapshot commit id), hy_validate) into device memory + hy_poslist_translate
        -> (RowID tensor, per-chunk region begins, counts tensor, total)"""
        torch = self.torch
        rows, n_chunks = max(1, column.rows), column.n_chunks
        regions = torch.empty((rows, 2), dtype=torch.int32, device=self.device)
        offsets = torch.empty(n_chunks + 1, dtype=torch.int64, device=self.device)
        counts = torch.zeros(max(1, n_chunks), dtype=torch.int32, device=self.device)
        result = abi.ScanResult()
        result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS | abi.SCAN_MATERIALIZE_ALL_MATCH
        result.matches, result.capacity = regions.data_ptr(), rows
        result.offsets, result.counts = offsets.data_ptr(), counts.data_ptr()
        if visibility is None:
            abi.check(self.lib.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result)))
        else:
            abi.check(self.lib.hy_validate(column.handle, visibility[0], visibility[1], 1, C.byref(result)))
        is_reference = isinstance(column, (DeviceReferenceColumn, DeviceChunkedReferenceColumn)) or getattr(getattr(column, "host", None), "is_reference", False)
        if layout == abi.POSLIST_CHUNK_REGIONS and visibility is None and not is_reference and _is_data_column(column):
            # a scan over a DATA table in the chunk-region layout already IS the output: chunk c's RowIDs of chunk c in its region
            return regions, offsets, counts, None
        out = torch.empty((rows, 2), dtype=torch.int32, device=self.device)
        written = C.c_uint64(0)
        abi.check(self.lib.hy_poslist_translate(column.handle, C.byref(result), layout, out.data_ptr(), rows, C.byref(written)))
        return out, offsets, counts, int(written.value)

    def scan(self, column, predicate):
        """RowIDs of the matching rows as ONE flat device PosList that references the DATA table: hy_table_scan writes its
        chunk regions to device memory, hy_poslist_translate packs them and -- when `column` is a reference column, i.e. the
        output of an earlier scan or join -- replaces each match by the RowID it stands for (table_scan.cpp:158-196).  Eight
        bytes (the match count) cross to the host; no PosList does."""
        out, _, _, total = self._device_scan(column, predicate, abi.POSLIST_DENSE)
        return out[:total]

    def scan_chunked(self, column, predicate):
        """The same scan with the reference's output shape kept: one PosList per input chunk (DevicePosLists), each still
        referencing the one data chunk its input chunk referenced.  The per-chunk counts (4 bytes per chunk) cross to the host."""
        out, offsets, counts, _ = self._device_scan(column, predicate, abi.POSLIST_CHUNK_REGIONS)
        begin = offsets[:column.n_chunks].cpu().numpy()
        count = counts[:column.n_chunks].cpu().numpy().astype(np.int64)
        if isinstance(column, DeviceChunkedReferenceColumn):
            base_chunk = column.pos.base_chunk[column.chunk_of_segment] if len(column.chunk_of_segment) else np.full(column.n_chunks, abi.INVALID_CHUNK_ID, dtype=np.int64)
        elif isinstance(column, DeviceReferenceColumn):
            base_chunk = np.full(column.n_chunks, abi.INVALID_CHUNK_ID, dtype=np.int64)
        else:
            base_chunk = np.arange(column.n_chunks, dtype=np.int64)                # a data table: chunk c's matches are rows of chunk c
        return DevicePosLists(out, begin, count, base_chunk)

    def validate_chunked(self, mvcc_column, our_tid, snapshot_commit_id):
        """Validate in front of the scans (what every SQL-driven plan has): the rows of the data table visible to
        the transaction, one PosList per chunk, on the device."""
        out, offsets, counts, _ = self._device_scan(mvcc_column, None, abi.POSLIST_CHUNK_REGIONS, visibility=(our_tid, snapshot_commit_id))
        begin = offsets[:mvcc_column.n_chunks].cpu().numpy()
        count = counts[:mvcc_column.n_chunks].cpu().numpy().astype(np.int64)
        return DevicePosLists(out, begin, count, np.arange(mvcc_column.n_chunks, dtype=np.int64))

    def reference_column_chunked(self, base, pos_lists):
        return DeviceChunkedReferenceColumn(self.lib, base, pos_lists)

    def reference_column(self, base, rows, chunk_rows):
        return DeviceReferenceColumn(self.lib, base, rows, chunk_rows)

    def projection(self, op, left, right):
        from .operators import projection_arithmetic
        return projection_arithmetic(op, left, right)

    def export(self, column, with_nulls=True):
        torch = self.torch
        values = torch.empty(column.rows, dtype=self._types[column.data_type], device=self.device)
        nulls = torch.zeros(column.rows, dtype=torch.uint8, device=self.device) if with_nulls else None
        abi.check(self.lib.hy_column_export(column.handle, values.data_ptr(), nulls.data_ptr() if nulls is not None else None))
        return values, nulls

    def value_column(self, values, chunk_rows, null_bytes=None):
        data_type = {v: k for k, v in self._types.items()}[values.dtype]
        return DeviceValueColumn(self.lib, values.contiguous(), chunk_rows, data_type, null_bytes)

    def join(self, left, right, mode):
        """-> (left positions [n, 2] int32, right positions [n, 2] int32 or None for semi / anti joins), device tensors"""
        torch = self.torch
        capacity = max(1, left.rows, right.rows)   # enough for a key / foreign-key join; otherwise the join says what it needs
        slice_capacity = max(left.rows, right.rows) // 131070 + max(left.n_chunks, right.n_chunks) + 600
        slice_offsets = torch.zeros(slice_capacity + 2, dtype=torch.int64, device=self.device)
        while True:
            from .operators import pair_lists
            left_pos, right_pos, _arena = pair_lists(torch, self.device, capacity)   # (one allocation, the lists 1.25 MiB apart modulo 2 MiB)
            r = abi.JoinResult()
            r.mem, r.radix_bits = abi.MEM_DEVICE, 0xFFFFFFFF
            r.left_pos, r.right_pos, r.capacity = left_pos.data_ptr(), right_pos.data_ptr(), capacity
            r.slice_offsets, r.slice_capacity = slice_offsets.data_ptr(), slice_capacity
            status = self.lib.hy_join_hash(left.handle, right.handle, mode, C.byref(r))
            if status == abi.ERR_CAPACITY and int(r.n_pairs) > capacity:   # (one pass 1 wasted: only joins that multiply rows)
                capacity = int(r.n_pairs)
                continue
            abi.check(status)
            break
        pairs = int(r.n_pairs)
        semi = mode in (abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_ANTI_NULL_AS_FALSE)
        return left_pos[:pairs], (None if semi else right_pos[:pairs])

    def repartition(self, column, parts, first_chunk):
        """-> (keys tensor, RowIDs [n, 2] int32, tuples per destination)"""
        torch = self.torch
        counts = (C.c_uint64 * parts)()
        keys = torch.empty(max(1, column.rows), dtype=self._types[column.data_type], device=self.device)
        rows = torch.empty((max(1, column.rows), 2), dtype=torch.int32, device=self.device)
        abi.check(self.lib.hy_repartition_pack(column.handle, parts, first_chunk, keys.data_ptr(), rows.data_ptr(), column.rows, counts))
        abi.check(self.lib.hy_synchronize())
        per = [int(c) for c in counts]
        total = sum(per)
        return keys[:total], rows[:total], per

    def gather_row_ids(self, table, chunk_rows, positions):
        torch = self.torch
        out = torch.empty_like(positions)
        if positions.shape[0]:
            table = table.contiguous()
            abi.check(self.lib.hy_gather_row_ids(table.data_ptr(), table.shape[0], chunk_rows, positions.contiguous().data_ptr(), positions.shape[0], out.data_ptr()))
        return out

    def synchronize(self):
        abi.check(self.lib.hy_synchronize())


# ---- AggregateHash ---------------------------------------------------------------------------------------------------
_MERGEABLE = (abi.AGG_MIN, abi.AGG_MAX, abi.AGG_SUM, abi.AGG_AVG, abi.AGG_COUNT, abi.AGG_ANY)
# STDDEV_SAMP travels as two cells: (SUM, COUNT) and (M2 = sum of squared deviations from the rank's mean, COUNT); two partials combine
# as  M2 = M2_a + M2_b + (mean_b - mean_a)^2 * n_a * n_b / (n_a + n_b)  (the pairwise update of Chan, Golub, LeVeque; the reference's own
# per-row update, abstract_aggregate_operator.hpp:83-113, is its n_b == 1 case).  COUNT(DISTINCT) has its own exchange (sharded_aggregate).
_MOMENT_SUM, _MOMENT_M2 = 1000, 1001
_INT_TYPES = (abi.TYPE_INT, abi.TYPE_LONG)


MAX_AGGREGATES_PER_CALL = 8   # hy_aggregate_hash's limit: device accumulators per call (STDDEV_SAMP takes two)


def _calls_of(plan):
    """The plan cut into executor calls of at most MAX_AGGREGATES_PER_CALL device accumulators each."""
    calls, weight = [[]], 0
    for function, column in plan:
        cost = 2 if function == abi.AGG_STDDEV_SAMP else 1
        if weight + cost > MAX_AGGREGATES_PER_CALL:
            calls.append([])
            weight = 0
        calls[-1].append((function, column))
        weight += cost
    return [call for call in calls if call]


def _local_partials(ex, groupby, aggregates):
    """The shard's groups: per group the key values, the first row and per aggregate (value, count of non-NULL inputs).
    The partial aggregates (each distinct (function, column) once, plus ANY of every GROUP BY column for the key values) run
    in as few executor calls as the per-call limit allows; the calls group the same rows in the same order."""
    plan, index = [], {}

    def want(function, column):
        key = (function, id(column))
        if key not in index:
            index[key] = len(plan)
            plan.append((function, column))
        return index[key]

    cells = []
    for function, column in aggregates:
        if function == _MOMENT_M2:     # (value = the rank's STDDEV_SAMP, turned into M2 below)
            cells.append((want(abi.AGG_STDDEV_SAMP, column), want(abi.AGG_COUNT, column)))
            continue
        if function == _MOMENT_SUM:
            function = abi.AGG_SUM
        if function not in _MERGEABLE:
            raise NotImplementedError(f"aggregate function {function} has no cross-rank merge rule in this helper (sharded_aggregate splits COUNT DISTINCT "
                                      "and STDDEV_SAMP into mergeable parts first)")
        cells.append((want(abi.AGG_SUM if function == abi.AGG_AVG else function, column), want(abi.AGG_COUNT, column)))
    key_cells = [want(abi.AGG_ANY, g) for g in groupby]
    shape = groupby[0] if groupby else next((c for _, c in aggregates if c is not None), None)
    if shape is None or ex.rows_of(shape) == 0:
        return [], [], []
    columns, first = [], None
    for call in _calls_of(plan):
        result = ex.aggregate(groupby, call)
        if first is None:
            first = result
        assert result.n_groups == first.n_groups
        columns += [result.column(i) for i in range(len(call))]
    n = first.n_groups
    keys = [tuple(columns[c][g] for c in key_cells) for g in range(n)]
    rows = [(int(first.row_ids[g][0]), int(first.row_ids[g][1])) for g in range(n)]
    values = [[(columns[v][g], columns[c][g]) for v, c in cells] for g in range(n)]
    for a, (function, _) in enumerate(aggregates):
        if function == _MOMENT_M2:   # s -> M2 = s^2 (n - 1); one row: 0
            for g in range(n):
                deviation, count = values[g][a]
                values[g][a] = (0.0 if deviation is None else float(deviation) ** 2 * (int(count) - 1), count)
    return keys, rows, values


def aggregate_groups(ex, groupby, aggregates):
    """One GPU: [(key tuple, [aggregate values])] in the reference's group order (the executor's)."""
    keys, _, values = _local_partials(ex, groupby, aggregates)
    out = []
    for key, row in zip(keys, values):
        cells = []
        for (function, _), (value, count) in zip(aggregates, row):
            if function == abi.AGG_COUNT:
                cells.append(count)
            elif value is None or not count:
                cells.append(None)
            elif function == abi.AGG_AVG:
                cells.append(float(value) / count)
            else:
                cells.append(value)
        out.append((key, cells))
    return out


def _aggregate_is_float(function, column):
    return column is not None and column.data_type in (abi.TYPE_FLOAT, abi.TYPE_DOUBLE) and function != abi.AGG_COUNT


def shared_long_strings(comm, local_strings):
    """The distinct strings of five or more bytes of a string GROUP BY column over ALL ranks, in one order on every rank (rank by
    rank, first appearance inside a rank): what string_keys.AggregateKeyNames(shared_long_strings=...) needs to hand out the same
    id for the same string everywhere.  local_strings: this rank's strings (any iterable of str / bytes; short ones are ignored)."""
    seen, mine = set(), []
    for value in local_strings:
        data = value if isinstance(value, bytes) else str(value).encode("utf-8")
        if len(data) >= 5 and data not in seen:
            seen.add(data)
            mine.append(data)
    gathered = [None] * comm.world
    comm.dist.all_gather_object(gathered, mine)
    out, seen = [], set()
    for part in gathered:
        for data in part:
            if data not in seen:
                seen.add(data)
                out.append(data)
    return out


def _check_key_names(key_names):
    """AggregateKeyNames of strings of five or more bytes are ids in order of first appearance (aggregate_hash.cpp:903-914): names
    inside one process.  Merging groups across ranks by key VALUE needs ids every rank agrees on."""
    for names in key_names or ():
        if names is not None and names.has_long_strings and not names.shared:
            raise NotImplementedError("GROUP BY a string column with entries of five or more bytes across ranks needs shared key names: "
                                      "string_keys.AggregateKeyNames(shared_long_strings=distributed.shared_long_strings(comm, strings))")


def sharded_aggregate(comm, ex, groupby, aggregates, first_chunk, total_rows_hint=None, key_names=None):
    """groupby / aggregates: THIS RANK's chunk range of the columns (executor columns); first_chunk: the range's first chunk
    id in the whole table.  Every rank returns the same list of (key tuple, [aggregate values]) in the reference's group
    order.  SUM / AVG over floating-point columns: double additions in a different order than the sequential reference
    (1e-9 relative, as on one GPU); everything else exact.  key_names: per GROUP BY column the string_keys.AggregateKeyNames its
    values came from, or None for numeric columns (string columns MUST be declared: see _check_key_names)."""
    _check_key_names(key_names)
    # STDDEV_SAMP -> (sum, count) + (M2, count); COUNT(DISTINCT x) -> its own exchange of the distinct (group, x) tuples
    internal, layout = [], []   # layout[a] = ("plain", index) | ("stddev", index of the sum cell) | ("distinct", column)
    for function, column in aggregates:
        if function == abi.AGG_STDDEV_SAMP:
            layout.append(("stddev", len(internal)))
            internal += [(_MOMENT_SUM, column), (_MOMENT_M2, column)]
        elif function == abi.AGG_COUNT_DISTINCT:
            layout.append(("distinct", column))
        else:
            layout.append(("plain", len(internal)))
            internal.append((function, column))
    keys, rows, values = _local_partials(ex, groupby, internal)
    shape = groupby[0] if groupby else next((c for _, c in aggregates if c is not None), None)
    local_rows = ex.rows_of(shape) if shape is not None else 0
    key_types = [g.data_type for g in groupby]
    merged = _merge_partials(comm, keys, rows, values, [f for f, _ in internal], [f in (_MOMENT_SUM, _MOMENT_M2) or _aggregate_is_float(f, c) for f, c in internal],
                             key_types, local_rows, first_chunk, raw_cells=True)
    distinct = {}
    for a, (kind, column) in enumerate(layout):
        if kind != "distinct":
            continue
        if len(groupby) + 1 > 16:
            raise NotImplementedError("COUNT(DISTINCT) across ranks groups by the GROUP BY columns and the counted column: at most seven GROUP BY columns")
        # the rank's distinct (group key, value) tuples = the groups of GROUP BY (keys..., value) (DISTINCT is a GROUP BY without aggregates,
        # aggregate_hash.cpp:1024-1061); merged across the ranks like any groups; NULL values are not counted
        tuple_keys, tuple_rows, tuple_values = _local_partials(ex, list(groupby) + [column], [(abi.AGG_COUNT, None)])
        tuples = _merge_partials(comm, tuple_keys, tuple_rows, tuple_values, [abi.AGG_COUNT], [False], key_types + [column.data_type], local_rows, first_chunk,
                                 force_general=True)
        tally = {}
        for key, _ in tuples:
            if key[-1] is not None:
                tally[key[:-1]] = tally.get(key[:-1], 0) + 1
        distinct[a] = tally
    out = []
    for key, cells in merged:
        row = []
        for a, (kind, where) in enumerate(layout):
            if kind == "distinct":
                row.append(distinct[a].get(key, 0))
            elif kind == "stddev":
                (total, count), (m2, _) = cells[where], cells[where + 1]
                row.append(None if count < 2 or m2 is None else float(np.sqrt(m2 / (count - 1))))
            else:
                value, count = cells[where]
                function = aggregates[a][0]
                if function == abi.AGG_COUNT:
                    row.append(count)
                elif count == 0 or value is None:
                    row.append(None)
                elif function == abi.AGG_AVG:
                    row.append(float(value) / count)
                else:
                    row.append(value)
        out.append((key, row))
    return out


def expression_type(tree):
    """Result type of an expression tree (operators.expression): expression_common_type, expression_utils.cpp:172-204."""
    if hasattr(tree, "data_type"):
        return tree.data_type
    if tree is None:
        return abi.TYPE_NULL
    if len(tree) == 2:
        return tree[0]
    left, right = expression_type(tree[1]), expression_type(tree[2])
    if left == abi.TYPE_NULL:
        return right
    if right == abi.TYPE_NULL:
        return left
    if abi.TYPE_DOUBLE in (left, right):
        return abi.TYPE_DOUBLE
    if abi.TYPE_LONG in (left, right):
        return abi.TYPE_DOUBLE if abi.TYPE_FLOAT in (left, right) else abi.TYPE_LONG
    return abi.TYPE_FLOAT if abi.TYPE_FLOAT in (left, right) else abi.TYPE_INT


def _tree_key(tree):
    if hasattr(tree, "data_type"):
        return ("column", id(tree))
    if tree is None or len(tree) == 2:
        return ("literal", tree)
    return (tree[0], _tree_key(tree[1]), _tree_key(tree[2]))


def sharded_scan_project_aggregate(comm, ex, filters, groupby, aggregates, first_chunk, key_names=None):
    """TableScan(s) -> Projection -> AggregateHash over a chunk-sharded table: every rank runs the fused pass
    (ex.scan_project_aggregate) over ITS chunks of the columns -- filters [(column, predicate)], groupby [column], aggregates
    [(MIN / MAX / SUM / AVG / COUNT, expression tree or None)] -- and the partial groups are merged like sharded_aggregate's
    (same exchange, same group order; the immediate-key shortcut is decided on the rows that passed the filters on all ranks).
    The key values travel as MIN(key column) of the group (all its rows hold the same value)."""
    _check_key_names(key_names)
    plan, index = [], {}

    def want(function, tree):
        key = (function, _tree_key(tree))
        if key not in index:
            index[key] = len(plan)
            plan.append((function, tree))
        return index[key]

    cells = []
    for function, tree in aggregates:
        if function not in (abi.AGG_MIN, abi.AGG_MAX, abi.AGG_SUM, abi.AGG_AVG, abi.AGG_COUNT):
            raise NotImplementedError(f"aggregate function {function} is not part of the fused pass: run the operator chain")
        cells.append((want(abi.AGG_SUM if function == abi.AGG_AVG else function, tree), want(abi.AGG_COUNT, tree)))
    key_cells = [want(abi.AGG_MIN, g) for g in groupby]
    passed_cell = want(abi.AGG_COUNT, None)
    columns, first = [], None
    for call in _calls_of(plan):   # (every call is a pass over the shard; they group the same rows in the same order)
        result = ex.scan_project_aggregate(filters, groupby, call)
        if first is None:
            first = result
        assert result.n_groups == first.n_groups
        columns += [result.column(i) if result.n_groups else [] for i in range(len(call))]
    n = first.n_groups
    if not groupby and n == 1 and not columns[passed_cell][0]:
        n = 0   # (no GROUP BY, nothing passed: the one row of NULLs / zero counts is produced after the merge, not by every rank)
    keys = [tuple(columns[c][g] for c in key_cells) for g in range(n)]
    rows = [(int(first.row_ids[g][0]), int(first.row_ids[g][1])) for g in range(n)]
    values = [[(columns[v][g], columns[c][g]) for v, c in cells] for g in range(n)]
    local_rows = sum(int(columns[passed_cell][g]) for g in range(n))
    is_float = [tree is not None and function != abi.AGG_COUNT and expression_type(tree) in (abi.TYPE_FLOAT, abi.TYPE_DOUBLE) for function, tree in aggregates]
    merged = _merge_partials(comm, keys, rows, values, [f for f, _ in aggregates], is_float, [g.data_type for g in groupby], local_rows, first_chunk)
    if not groupby and not merged:
        merged = [((), [0 if f == abi.AGG_COUNT else None for f, _ in aggregates])]
    return merged


def _merge_partials(comm, keys, rows, values, functions, is_float_aggregate, key_types, local_rows, first_chunk, raw_cells=False, force_general=False):
    """The exchange and the merge behind sharded_aggregate / sharded_scan_project_aggregate.  keys / rows / values: this rank's groups (key
    tuple, first row as (chunk, offset) of the shard, per aggregate (value, count of non-NULL inputs)); local_rows: the rows of the
    aggregate's input on this rank."""
    torch = comm.torch
    n_aggregates, n_keys = len(functions), len(key_types)
    device = comm._device
    first_rows = [((chunk + first_chunk) << 32) | offset for chunk, offset in rows]

    # ---- key ranges (integer GROUP BY columns only): decide between fixed slots and the general merge, on every rank alike
    integer_keys = all(t in _INT_TYPES for t in key_types)
    BIG = 1 << 62
    low = torch.full((max(1, n_keys),), BIG, dtype=torch.int64, device=device)
    high = torch.full((max(1, n_keys),), -BIG, dtype=torch.int64, device=device)
    if integer_keys and keys:
        for k in range(n_keys):
            present = [key[k] for key in keys if key[k] is not None]
            if present:
                low[k], high[k] = min(present), max(present)
    totals = torch.tensor([local_rows], dtype=torch.int64, device=device)
    comm.all_reduce(low, "min")
    comm.all_reduce(high, "max")
    comm.all_reduce(totals, "sum")
    total_rows = int(totals.item())
    low_l, high_l = [int(x) for x in low.cpu().tolist()], [int(x) for x in high.cpu().tolist()]
    spans = [max(0, h - l + 1) + 1 for l, h in zip(low_l, high_l)][:n_keys]   # + 1: slot 0 of every column is NULL
    slots = 1
    for s in spans:
        slots *= s
    fixed = integer_keys and slots <= FIXED_SLOT_LIMIT and not force_general and _MOMENT_M2 not in functions   # (moments merge pairwise: the general path)

    merged = {}   # key tuple -> [first row, last row, [[value, count], ...]]
    if fixed:
        def slot_of(key):
            index = 0
            for k in range(n_keys):
                index = index * spans[k] + (0 if key[k] is None else int(key[k]) - low_l[k] + 1)
            return index

        isum = torch.zeros((slots, 2 * n_aggregates + 1), dtype=torch.int64, device="cpu")     # integer SUMs | counts | group present
        fsum = torch.zeros((slots, max(1, n_aggregates)), dtype=torch.float64, device="cpu")
        imin = torch.full((slots, n_aggregates + 1), BIG, dtype=torch.int64, device="cpu")      # integer MINs | first row
        imax = torch.full((slots, n_aggregates + 1), -BIG, dtype=torch.int64, device="cpu")     # integer MAXs | last row
        fmin = torch.full((slots, max(1, n_aggregates)), float("inf"), dtype=torch.float64, device="cpu")
        fmax = torch.full((slots, max(1, n_aggregates)), float("-inf"), dtype=torch.float64, device="cpu")
        for key, first, row in zip(keys, first_rows, values):
            s = slot_of(key)
            isum[s, 2 * n_aggregates] = 1
            imin[s, n_aggregates] = first
            imax[s, n_aggregates] = first
            for a, (value, count) in enumerate(row):
                count = 0 if count is None else int(count)
                isum[s, n_aggregates + a] = count
                if value is None or (count == 0 and functions[a] != abi.AGG_COUNT):
                    continue
                is_float = is_float_aggregate[a]
                if functions[a] in (abi.AGG_SUM, abi.AGG_AVG):
                    if is_float:
                        fsum[s, a] = float(value)
                    else:
                        isum[s, a] = int(value)
                elif functions[a] in (abi.AGG_MIN, abi.AGG_ANY):
                    if is_float:
                        fmin[s, a] = float(value)
                    else:
                        imin[s, a] = int(value)
                elif functions[a] == abi.AGG_MAX:
                    if is_float:
                        fmax[s, a] = float(value)
                    else:
                        imax[s, a] = int(value)
        tensors = [(isum, "sum"), (fsum, "sum"), (imin, "min"), (fmin, "min"), (imax, "max"), (fmax, "max")]
        on_device = [(t.to(device), op) for t, op in tensors]
        for t, op in on_device:   # the exchange: all-reduce over RCCL, device buffers
            comm.all_reduce(t, op)
        isum, fsum, imin, fmin, imax, fmax = [t.cpu() for t, _ in on_device]
        for s in torch.nonzero(isum[:, 2 * n_aggregates]).flatten().tolist():
            key, rest = [], s
            for k in reversed(range(n_keys)):
                digit = rest % spans[k]
                rest //= spans[k]
                key.append(None if digit == 0 else digit - 1 + low_l[k])
            key = tuple(reversed(key))
            row = []
            for a in range(n_aggregates):
                count = int(isum[s, n_aggregates + a])
                is_float = is_float_aggregate[a]
                if functions[a] in (abi.AGG_SUM, abi.AGG_AVG):
                    value = float(fsum[s, a]) if is_float else int(isum[s, a])
                elif functions[a] in (abi.AGG_MIN, abi.AGG_ANY):
                    value = float(fmin[s, a]) if is_float else int(imin[s, a])
                elif functions[a] == abi.AGG_MAX:
                    value = float(fmax[s, a]) if is_float else int(imax[s, a])
                else:
                    value = 0
                row.append([value, count])
            merged[key] = [int(imin[s, n_aggregates]), int(imax[s, n_aggregates]), row]
    else:
        # general merge: every rank's groups, all-gathered as arrays (keys as doubles' / integers' bits in int64 + NULL flags)
        def bits(value, key_type):
            if value is None:
                return 0
            if key_type in (abi.TYPE_FLOAT, abi.TYPE_DOUBLE):
                return int(np.float64(value).view(np.int64))
            return int(value)

        g = len(keys)
        table = np.zeros((g, 2 * n_keys + 1 + 3 * n_aggregates), dtype=np.int64)
        for i, (key, first, row) in enumerate(zip(keys, first_rows, values)):
            for k in range(n_keys):
                table[i, 2 * k] = bits(key[k], key_types[k])
                table[i, 2 * k + 1] = 1 if key[k] is None else 0
            table[i, 2 * n_keys] = first
            for a, (value, count) in enumerate(row):
                base = 2 * n_keys + 1 + 3 * a
                table[i, base] = 0 if count is None else int(count)
                table[i, base + 1] = 0 if value is None else 1
                if value is not None:
                    is_float = is_float_aggregate[a]
                    table[i, base + 2] = int(np.float64(value).view(np.int64)) if is_float else int(value)
        parts = comm.all_gather_var(torch.from_numpy(table).to(device))
        for part in parts:   # rank order: deterministic floating-point sums
            for line in part.cpu().numpy():
                key = []
                for k in range(n_keys):
                    if line[2 * k + 1]:
                        key.append(None)
                    elif key_types[k] in (abi.TYPE_FLOAT, abi.TYPE_DOUBLE):
                        key.append(float(np.int64(line[2 * k]).view(np.float64)))
                    else:
                        key.append(int(line[2 * k]))
                key = tuple(key)
                first = int(line[2 * n_keys])
                entry = merged.get(key)
                if entry is None:
                    entry = merged[key] = [first, first, [[None, 0] for _ in range(n_aggregates)]]
                entry[0], entry[1] = min(entry[0], first), max(entry[1], first)
                for a in reversed(range(n_aggregates)):   # (descending: an M2 cell sits behind its sum cell and needs that cell's OLD value)
                    base = 2 * n_keys + 1 + 3 * a
                    count, has_value = int(line[base]), bool(line[base + 1])
                    is_float = is_float_aggregate[a]
                    value = (float(np.int64(line[base + 2]).view(np.float64)) if is_float else int(line[base + 2])) if has_value else None
                    cur = entry[2][a]
                    if functions[a] == _MOMENT_M2 and count:
                        # pairwise update: (n_a, sum_a, M2_a) so far, (n_b, sum_b, M2_b) arriving; the sum cell is a - 1
                        sum_base = 2 * n_keys + 1 + 3 * (a - 1)
                        n_b, sum_b = count, float(np.int64(line[sum_base + 2]).view(np.float64)) if line[sum_base + 1] else 0.0
                        n_a, sum_a = entry[2][a - 1][1], entry[2][a - 1][0] or 0.0
                        m2 = (cur[0] or 0.0) + (value or 0.0)
                        if n_a and n_b:
                            delta = sum_b / n_b - sum_a / n_a
                            m2 += delta * delta * n_a * n_b / (n_a + n_b)
                        cur[0], cur[1] = m2, cur[1] + count
                        continue
                    cur[1] += count
                    if value is None:
                        continue
                    if cur[0] is None:
                        cur[0] = value
                    elif functions[a] in (abi.AGG_SUM, abi.AGG_AVG, _MOMENT_SUM):
                        cur[0] = cur[0] + value
                    elif functions[a] == abi.AGG_MIN:
                        cur[0] = min(cur[0], value)
                    elif functions[a] == abi.AGG_MAX:
                        cur[0] = max(cur[0], value)
    # ---- the reference's group order: ascending key with NULL first under the immediate-key shortcut (one int32 GROUP BY
    #      column whose key range is below 1.2 x rows, aggregate_hash.cpp:770-804), else first occurrence (:388-401)
    immediate = False
    if n_keys == 1 and key_types[0] == abi.TYPE_INT and merged:
        present = [key[0] for key in merged if key[0] is not None]
        if present and (max(present) - min(present)) < total_rows * 1.2:
            immediate = True
    if immediate:
        order = sorted(merged.items(), key=lambda kv: (kv[0][0] is not None, kv[0][0] if kv[0][0] is not None else 0))
    else:
        order = sorted(merged.items(), key=lambda kv: kv[1][0])
    if raw_cells:
        return [(key, [tuple(cell) for cell in row]) for key, (first, last, row) in order]
    out = []
    for key, (first, last, row) in order:
        cells = []
        for a, function in enumerate(functions):
            value, count = row[a]
            if function == abi.AGG_COUNT:
                cells.append(count)
            elif count == 0 or value is None:
                cells.append(None)
            elif function == abi.AGG_AVG:
                cells.append(float(value) / count)
            else:
                cells.append(value)
        out.append((key, cells))
    return out


# ---- JoinHash --------------------------------------------------------------------------------------------------------
def _offset_chunks(torch, positions, first_chunk):
    """RowIDs [n, 2] int32 of a shard -> of the whole table (NULL_ROW_ID stays)."""
    if positions is None or first_chunk == 0 or positions.shape[0] == 0:
        return positions
    out = positions.clone()
    valid = out[:, 1] != _NULL_ROW
    out[:, 0] = torch.where(valid, out[:, 0] + first_chunk, out[:, 0])
    return out


def sharded_join_broadcast(comm, ex, build, probe, mode, first_probe_chunk, build_chunk_rows, build_is_left=True):
    """Broadcast-build: `build` is this rank's chunk range of the build table's join column, `probe` of the probe table's.
    Returns this rank's (build RowIDs, probe RowIDs) as device tensors [n, 2] int32, RowIDs of the WHOLE tables.  The shards
    of the build column must be whole chunks of `build_chunk_rows` rows (all but the table's last): gathered in rank order
    they are the table."""
    # hy_join_hash picks the build side by mode (join_hash.cpp:139-155): Left, Semi and Anti* build on the right input, Right on the
    # left one.  The gathered table must be THAT side -- as the probe or outer side it would be probed / emitted by every rank, and
    # the union over the ranks would hold its unmatched (Left, Anti*) or matched (Semi) rows once per rank.
    allowed = {abi.JOIN_INNER: (True, False), abi.JOIN_LEFT: (False,), abi.JOIN_SEMI: (False,), abi.JOIN_ANTI_NULL_AS_TRUE: (False,),
               abi.JOIN_ANTI_NULL_AS_FALSE: (False,), abi.JOIN_RIGHT: (True,)}
    if build_is_left not in allowed.get(mode, ()):
        raise NotImplementedError(f"broadcast-build join: mode {mode} builds on the {'right' if mode != abi.JOIN_RIGHT else 'left'} input -- "
                                  f"the gathered column must be that side (build_is_left={not build_is_left})")
    torch = comm.torch
    values, nulls = ex.export(build)
    gathered = torch.cat(comm.all_gather_var(values))
    gathered_nulls = torch.cat(comm.all_gather_var(nulls)) if nulls is not None else None
    whole = ex.value_column(gathered, build_chunk_rows, gathered_nulls)
    if build_is_left:
        left_pos, right_pos = ex.join(whole, probe, mode)
        return left_pos, _offset_chunks(torch, right_pos, first_probe_chunk)
    left_pos, right_pos = ex.join(probe, whole, mode)
    return right_pos, _offset_chunks(torch, left_pos, first_probe_chunk)


def sharded_join_repartition(comm, ex, left, right, first_left_chunk, first_right_chunk, mode=abi.JOIN_INNER):
    """Hash repartition: both sides' (key, RowID) tuples go to rank  key % G  (one all-to-all per side), every rank joins
    what it received with `mode`.  NULL keys are not sent -- a NULL meets no partner anywhere -- so the modes that keep rows with
    NULL keys (join_hash.cpp:284-286) get them from the rank that holds them:
      Left / Right         the outer side's NULL-key rows leave with a NULL partner, once, from their own rank
      AntiNullAsFalse      the left side's NULL-key rows are part of the result
      AntiNullAsTrue       NULL = x is not false: a NULL key on the RIGHT side empties the result (join_hash.cpp:483-494), a NULL key on
                           the left side is kept only when the right table has no rows at all (ranks agree through two all-reduces)
    Returns (left RowIDs, right RowIDs or None) of the whole tables, device tensors; rows of different ranks in rank order."""
    torch = comm.torch
    semi = mode in (abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_ANTI_NULL_AS_FALSE)
    received = []
    for column, first_chunk in ((left, first_left_chunk), (right, first_right_chunk)):
        keys, rows, counts = ex.repartition(column, comm.world, first_chunk)
        recv_keys, _ = comm.all_to_all_var(keys, counts)
        recv_rows, _ = comm.all_to_all_var(rows, counts)
        received.append((recv_keys, recv_rows))
    (left_keys, left_rows), (right_keys, right_rows) = received
    left_column = ex.value_column(left_keys, REPARTITION_CHUNK)
    right_column = ex.value_column(right_keys, REPARTITION_CHUNK)
    left_pos, right_pos = ex.join(left_column, right_column, mode)
    out_left = ex.gather_row_ids(left_rows, REPARTITION_CHUNK, left_pos)
    out_right = ex.gather_row_ids(right_rows, REPARTITION_CHUNK, right_pos) if right_pos is not None else None
    if mode in (abi.JOIN_INNER, abi.JOIN_SEMI):
        return out_left, out_right

    from .operators import make_predicate
    def null_key_rows(column, first_chunk):
        rows = ex.scan(column, make_predicate(abi.PRED_IS_NULL, column.data_type))
        return _offset_chunks(torch, rows.to(out_left.device), first_chunk)

    null_partner = lambda n: torch.full((n, 2), -1, dtype=torch.int32, device=out_left.device)   # NULL_ROW_ID
    if mode == abi.JOIN_LEFT:
        mine = null_key_rows(left, first_left_chunk)
        return torch.cat([out_left, mine]), torch.cat([out_right, null_partner(mine.shape[0])])
    if mode == abi.JOIN_RIGHT:
        mine = null_key_rows(right, first_right_chunk)
        return torch.cat([out_left, null_partner(mine.shape[0])]), torch.cat([out_right, mine])
    mine = null_key_rows(left, first_left_chunk)
    if mode == abi.JOIN_ANTI_NULL_AS_FALSE:
        return torch.cat([out_left, mine]), None
    # AntiNullAsTrue
    flags = torch.tensor([ex.rows_of(right), int(null_key_rows(right, first_right_chunk).shape[0])], dtype=torch.int64, device=comm._device)
    comm.all_reduce(flags, "sum")
    right_rows_total, right_nulls_total = int(flags[0].item()), int(flags[1].item())
    if right_nulls_total:
        return out_left[:0], None
    if right_rows_total == 0:
        return torch.cat([out_left, mine]), None
    return out_left, None


# ---- bench.py --gpus N: the legs beside the weak-scaling scan ---------------------------------------------------------
def bench_legs(lib, torch, dist, device, rank, world, share_gpu, steps=5):
    """Strong-scaling scan, sharded Q1-core aggregate and both sharded joins on ONE SF10 table split over the ranks.
    Returns a dict on rank 0 (None elsewhere): every time is the maximum over the ranks."""
    import time
    from . import tpch, storage
    from .operators import make_predicate
    from .storage import DeviceColumn
    comm = Comm(dist).bind(torch.device("cpu") if share_gpu else device)
    ex = HipExecutor(device)
    data = tpch.TpchData(scale_factor=10.0, seed=42)   # the same table on every rank; each keeps its chunk range

    def timed(run, repeats=steps):
        run()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(repeats):
            run()
        torch.cuda.synchronize()
        elapsed = torch.tensor([(time.perf_counter() - t0) / repeats], dtype=torch.float64, device=comm._device)
        comm.all_reduce(elapsed, "max")
        return float(elapsed.item())

    out = {}
    # -- scan, strong scaling: one SF10 l_shipdate column, chunk ranges per rank, no collective in the data path
    days, whole = tpch.shipdate_column(tpch.LINEITEM_ROWS_SF10, seed=42)
    shard, first_chunk = shard_column(whole, world, rank)
    column = DeviceColumn(shard)
    matches = torch.empty((max(1, shard.rows), 2), dtype=torch.int32, device=device)
    offsets = torch.zeros(shard.n_chunks + 1, dtype=torch.int64, device=device)
    counts = torch.zeros(max(1, shard.n_chunks), dtype=torch.int32, device=device)
    result = abi.ScanResult()
    result.mem, result.flags = abi.MEM_DEVICE, abi.SCAN_CHUNK_REGIONS
    result.matches, result.capacity, result.offsets, result.counts = matches.data_ptr(), shard.rows, offsets.data_ptr(), counts.data_ptr()
    predicate = make_predicate(abi.PRED_LESS_THAN, abi.TYPE_INT, tpch.DAY_1995_01_01)
    seconds = timed(lambda: abi.check(lib.hy_table_scan(column.handle, C.byref(predicate), None, 0, C.byref(result))), repeats=max(steps, 20))
    found = torch.tensor([int(counts.sum().item())], dtype=torch.int64, device=comm._device)
    comm.all_reduce(found, "sum")
    out["scan_strong"] = {"rows": whole.rows, "ms": seconds * 1e3, "rows_per_s": whole.rows / seconds, "matches": int(found.item()),
                          "expected_matches": int((days < tpch.DAY_1995_01_01).sum())}
    del column, matches
    # -- aggregate: Q1 core as specified, per-rank partials + fixed-slot all-reduce
    groupby_host, measures_host, _ = tpch.q1_core_columns(data)
    groupby = [DeviceColumn(shard_column(c, world, rank)[0]) for c in groupby_host]
    measures = {name: DeviceColumn(shard_column(c, world, rank)[0]) for name, c in measures_host.items()}
    first_chunk = shard_column(groupby_host[0], world, rank)[1]
    aggregates = [(abi.AGG_SUM, measures["l_quantity"]), (abi.AGG_SUM, measures["l_extendedprice"]), (abi.AGG_AVG, measures["l_quantity"]),
                  (abi.AGG_AVG, measures["l_extendedprice"]), (abi.AGG_AVG, measures["l_discount"]), (abi.AGG_COUNT, None)]
    holder = {}

    def run_aggregate():
        holder["groups"] = sharded_aggregate(comm, ex, groupby, aggregates, first_chunk)

    seconds = timed(run_aggregate)
    out["aggregate_q1"] = {"rows": data.n_lineitems, "ms": seconds * 1e3, "rows_per_s": data.n_lineitems / seconds, "groups": len(holder["groups"]),
                           "count_star_total": int(sum(cells[5] for _, cells in holder["groups"])), "exchange": "fixed-slot all-reduce (sum / min / max)"}
    del groupby, measures
    # -- joins: orders x lineitem on the order key, both tables chunk-sharded
    orders_host = storage.make_column(data.o_orderkey, None, abi.ENC_UNENCODED)
    lineitem_host = storage.make_column(data.l_orderkey, None, abi.ENC_FRAME_OF_REFERENCE)
    orders_shard, first_orders = shard_column(orders_host, world, rank)
    lineitem_shard, first_lineitem = shard_column(lineitem_host, world, rank)
    orders, lineitem = DeviceColumn(orders_shard), DeviceColumn(lineitem_shard)
    total = data.n_orders + data.n_lineitems
    pairs = {}

    def run_broadcast():
        build_pos, probe_pos = sharded_join_broadcast(comm, ex, orders, lineitem, abi.JOIN_INNER, first_lineitem, abi.CHUNK_DEFAULT_SIZE)
        pairs["broadcast"] = build_pos.shape[0]

    def run_repartition():
        left_pos, right_pos = sharded_join_repartition(comm, ex, orders, lineitem, first_orders, first_lineitem)
        pairs["repartition"] = left_pos.shape[0]

    for name, run in (("join_broadcast_build", run_broadcast), ("join_repartition", run_repartition)):
        seconds = timed(run, repeats=min(steps, 3))
        n_pairs = torch.tensor([pairs["broadcast" if "broadcast" in name else "repartition"]], dtype=torch.int64, device=comm._device)
        comm.all_reduce(n_pairs, "sum")
        out[name] = {"rows": total, "ms": seconds * 1e3, "rows_per_s": total / seconds, "pairs": int(n_pairs.item()), "expected_pairs": data.n_lineitems}
    return out if rank == 0 else None
