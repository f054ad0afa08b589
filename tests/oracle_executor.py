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


class OracleExecutor:
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
        result = oracle_scan(column, predicate, flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
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
        cells = column_values(column)
        nulls = np.array([c is None for c in cells], dtype=np.uint8)
        values = np.array([0 if c is None else c for c in cells], dtype=_NP[column.data_type])
        return torch.from_numpy(values), (torch.from_numpy(nulls) if with_nulls else None)

    def value_column(self, values, chunk_rows, null_bytes=None):
        nulls = null_bytes.numpy().astype(bool) if null_bytes is not None and bool(null_bytes.any()) else None
        return build_column(values.numpy(), nulls, chunk_rows, abi.ENC_UNENCODED)

    def join(self, left, right, mode):
        result = oracle_join(left, right, mode)
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
