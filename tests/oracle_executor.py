"""The per-rank executor of the CPU tests: the same interface as hyrise_amd.distributed.HipExecutor, computed by the CPU
oracle and numpy on host tensors (test infrastructure: there is no GPU in the CPU test environment)."""
import numpy as np
import torch

from hyrise_amd import abi
from support import build_column, column_values, oracle_aggregate, oracle_join

_NP = {abi.TYPE_INT: np.int32, abi.TYPE_LONG: np.int64, abi.TYPE_FLOAT: np.float32, abi.TYPE_DOUBLE: np.float64}


class _HostPosLists:
    def __init__(self, lists, base_chunk):
        self.lists, self.base_chunk = lists, base_chunk
        self.total = sum(len(rows) for rows in lists)


def _null_bits(segment, n):
    if segment.nulls is None:
        return np.zeros(n, dtype=bool)
    return np.unpackbits(np.ascontiguousarray(segment.nulls).view(np.uint8), bitorder="little")[:n].astype(bool)


def _decode_data_column(column):
    """(values, NULL flags) of a DATA column as flat numpy arrays (Unencoded / Dictionary / FrameOfReference segments; None otherwise);
    remembered on the column object (columns do not change)."""
    cached = getattr(column, "_decoded_flat", None)
    if cached is not None:
        return cached
    values, nulls = [], []
    for segment in column.segments:
        n = segment.size
        if segment.encoding == abi.ENC_DICTIONARY and getattr(segment, "bits", 0) == 0:
            ids = np.asarray(segment.data[:n]).astype(np.int64)
            is_null = ids >= segment.aux_size
            dictionary = np.asarray(segment.aux) if segment.aux_size else np.zeros(1, dtype=_NP[column.data_type])
            values.append(dictionary[np.where(is_null, 0, ids)])
            nulls.append(is_null)
        elif segment.encoding == abi.ENC_FRAME_OF_REFERENCE and getattr(segment, "bits", 0) == 0:
            raw = np.asarray(segment.data[:n]).astype(np.int64)
            minima = np.asarray(segment.aux).astype(np.int64)[np.arange(n) // abi.FOR_BLOCK_SIZE] if n else np.zeros(0, dtype=np.int64)
            values.append((raw + minima).astype(np.int32))
            nulls.append(_null_bits(segment, n))
        elif segment.encoding == abi.ENC_UNENCODED:
            values.append(np.asarray(segment.data[:n]))
            nulls.append(_null_bits(segment, n))
        else:
            return None
    flat = (np.concatenate(values) if values else np.zeros(0, dtype=_NP[column.data_type]), np.concatenate(nulls) if nulls else np.zeros(0, dtype=bool))
    try:
        column._decoded_flat = flat
    except AttributeError:
        pass
    return flat


def _decode_column(column):
    """... of a data column or of a reference column over one (numpy gathers instead of a Python loop per cell: the SSB plans export
    millions of cells per query)."""
    if not column.segments or column.segments[0].encoding != abi.ENC_REFERENCE:
        return _decode_data_column(column)
    base = column.segments[0].ref
    decoded = _decode_data_column(base)
    if decoded is None or any(s.encoding != abi.ENC_REFERENCE or s.ref is not base for s in column.segments):
        return None
    begins = np.concatenate([[0], np.cumsum([s.size for s in base.segments])]).astype(np.int64)
    rows = []
    for segment in column.segments:
        if segment.data is None:
            rows.append(np.stack([np.full(segment.size, segment.ref_chunk_id, dtype=np.uint32), np.arange(segment.size, dtype=np.uint32)], axis=1))
        else:
            rows.append(np.asarray(segment.data).reshape(-1, 2)[:segment.size].astype(np.uint32))
    rows = np.concatenate(rows) if rows else np.zeros((0, 2), dtype=np.uint32)
    null_row = rows[:, 1] == 0xFFFFFFFF
    flat = np.where(null_row, 0, begins[np.where(null_row, 0, rows[:, 0]).astype(np.int64)] + rows[:, 1].astype(np.int64))
    if len(decoded[0]) == 0:
        return np.zeros(len(rows), dtype=_NP[column.data_type]), np.ones(len(rows), dtype=bool)
    return decoded[0][flat], decoded[1][flat] | null_row


class OracleExecutor:
    def __init__(self, threads=1):
        self.threads = threads   # of the oracle's scan and join (chunk ranges / radix partitions per thread); the aggregate is sequential like the reference's

    def column(self, host_column):
        return host_column

    def rows_of(self, column):
        return column.rows

    def aggregate(self, groupby, aggregates):
        return oracle_aggregate(groupby, aggregates)

    def scan_project_aggregate(self, filters, groupby, aggregates):
        """The fused pass, run the reference's way (support.oracle_chain: scan -> scan -> ... -> arithmetic node by node -> aggregate);
        the groups' representative rows are translated to rows of the data table, which is what hy_scan_project_aggregate returns."""
        from support import oracle_chain
        result, base_rows, sizes = oracle_chain(filters, groupby, aggregates)
        n = result.n_groups
        if n and len(base_rows):
            first_of_chunk = np.concatenate([[0], np.cumsum(sizes)[:-1]])
            flat = first_of_chunk[result.row_ids[:n, 0].astype(np.int64)] + result.row_ids[:n, 1].astype(np.int64)
            result.row_ids[:n] = base_rows[flat]
        return result

    def scan(self, column, predicate):
        from support import oracle_scan
        result = oracle_scan(column, predicate, flags=abi.SCAN_MATERIALIZE_ALL_MATCH, threads=self.threads)
        matches = result.matches[:result.total].copy()
        if column.segments and column.segments[0].encoding == abi.ENC_REFERENCE:   # (c, o) -> the RowID at position o of chunk c's PosList
            begins = np.concatenate([[0], np.cumsum([s.size for s in column.segments])])
            pos = np.concatenate([np.asarray(s.data).reshape(-1, 2) for s in column.segments]) if column.rows else matches[:0]
            matches = pos[begins[matches[:, 0]] + matches[:, 1]].astype(np.uint32)
        return torch.from_numpy(np.ascontiguousarray(matches).view(np.int32).copy())

    def scan_chunked(self, column, predicate):
        """-> per input chunk the matching RowIDs of the DATA table (host arrays) + the one data chunk each list references"""
        from support import oracle_scan
        result = oracle_scan(column, predicate, flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
        lists, base_chunk = [], []
        for c, segment in enumerate(column.segments):
            matches = result.pos_list(c).copy()
            if segment.encoding == abi.ENC_REFERENCE:
                matches = np.asarray(segment.data, dtype=np.uint32).reshape(-1, 2)[matches[:, 1]] if segment.data is not None else \
                    np.stack([np.full(len(matches), segment.ref_chunk_id, dtype=np.uint32), matches[:, 1]], axis=1)
                base_chunk.append(segment.ref_chunk_id)
            else:
                base_chunk.append(c)
            lists.append(matches)
        return _HostPosLists(lists, base_chunk)

    def validate_chunked(self, mvcc_column, our_tid, snapshot_commit_id):
        from support import oracle_validate
        result = oracle_validate(mvcc_column, our_tid, snapshot_commit_id, flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
        return _HostPosLists([result.pos_list(c).copy() for c in range(mvcc_column.n_chunks)], list(range(mvcc_column.n_chunks)))

    def reference_column_chunked(self, base, pos_lists):
        from hyrise_amd import storage
        keep = [c for c, rows in enumerate(pos_lists.lists) if len(rows)]
        if not keep:
            return storage.make_reference_column(base, [np.zeros((0, 2), dtype=np.uint32)], [None])
        return storage.make_reference_column(base, [pos_lists.lists[c] for c in keep],
                                             [None if pos_lists.base_chunk[c] == abi.INVALID_CHUNK_ID else pos_lists.base_chunk[c] for c in keep])

    def reference_column(self, base, rows, chunk_rows):
        from hyrise_amd import storage
        pos = rows.numpy().view(np.uint32).reshape(-1, 2)
        chunks = [pos[begin:begin + chunk_rows] for begin in range(0, len(pos), chunk_rows)] or [pos[:0]]
        return storage.make_reference_column(base, chunks)

    def projection(self, op, left, right):
        """left <op> right, operands: columns or literals ((HY_TYPE_*, value) / None), with the chunk layout of the column operand(s)."""
        from hyrise_amd import storage
        from support import oracle_arithmetic

        def operand(x):
            if x is None or not hasattr(x, "segments"):
                return x
            cells = column_values(x)
            nulls = np.array([c is None for c in cells], dtype=bool)
            return np.array([0 if c is None else c for c in cells], dtype=_NP[x.data_type]), (nulls if nulls.any() else None)

        shape = left if hasattr(left, "segments") else right
        values, nulls = oracle_arithmetic(op, operand(left), operand(right), n=shape.rows)
        segments, begin = [], 0
        for segment in shape.segments:
            segments.append(storage.encode_segment(values[begin:begin + segment.size], nulls[begin:begin + segment.size] if nulls.any() else None, abi.ENC_UNENCODED))
            begin += segment.size
        return storage.HostColumn(segments, storage.TYPE_OF_NP[values.dtype])

    def export(self, column, with_nulls=True):
        decoded = _decode_column(column)
        if decoded is None:   # (layouts the vectorised decoder below does not cover: cell by cell)
            cells = column_values(column)
            nulls = np.array([c is None for c in cells], dtype=np.uint8)
            values = np.array([0 if c is None else c for c in cells], dtype=_NP[column.data_type])
        else:
            values, is_null = decoded
            values = np.where(is_null, np.zeros(1, dtype=values.dtype), values).astype(_NP[column.data_type])
            nulls = is_null.astype(np.uint8)
        return torch.from_numpy(np.ascontiguousarray(values)), (torch.from_numpy(nulls) if with_nulls else None)

    def value_column(self, values, chunk_rows, null_bytes=None):
        nulls = null_bytes.numpy().astype(bool) if null_bytes is not None and bool(null_bytes.any()) else None
        return build_column(values.numpy(), nulls, chunk_rows, abi.ENC_UNENCODED)

    def join(self, left, right, mode):
        result = oracle_join(left, right, mode, threads=self.threads)
        n = result.n_pairs
        semi = mode in (abi.JOIN_SEMI, abi.JOIN_ANTI_NULL_AS_TRUE, abi.JOIN_ANTI_NULL_AS_FALSE)
        left_pos = torch.from_numpy(result.left[:n].astype(np.int64).astype(np.uint32).view(np.int32).copy())
        right_pos = None if semi else torch.from_numpy(result.right[:n].view(np.int32).copy())
        return left_pos, right_pos

    def repartition(self, column, parts, first_chunk):
        cells = column_values(column)
        rows = []
        for c, seg in enumerate(column.segments):
            rows.extend((c + first_chunk, i) for i in range(seg.size))
        keep = [i for i, cell in enumerate(cells) if cell is not None]
        keys = np.array([cells[i] for i in keep], dtype=_NP[column.data_type])
        row_ids = np.array([rows[i] for i in keep], dtype=np.uint32).reshape(-1, 2)
        dest = (keys.astype(np.int64).view(np.uint64) % np.uint64(parts)).astype(np.int64)
        order = np.argsort(dest, kind="stable")
        counts = [int((dest == p).sum()) for p in range(parts)]
        return torch.from_numpy(keys[order].copy()), torch.from_numpy(row_ids[order].view(np.int32).copy()), counts

    def gather_row_ids(self, table, chunk_rows, positions):
        t = table.numpy().view(np.uint32)
        p = positions.numpy().view(np.uint32)
        out = np.full_like(p, 0xFFFFFFFF)
        valid = p[:, 1] != 0xFFFFFFFF
        flat = p[valid, 0].astype(np.int64) * chunk_rows + p[valid, 1].astype(np.int64)
        out[valid] = t[flat]
        return torch.from_numpy(out.view(np.int32).copy())

    def synchronize(self):
        pass
