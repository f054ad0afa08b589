"""Pins the CPU restatement of Validate (oracle/validate.c) against the reference's own known answers: the MVCC truth
table of operators/validate_visibility_test.cpp:45-131 (our_tid = 2, snapshot = 2) and the four _is_entire_chunk_visible cases of
operators/validate_test.cpp:163-214, and states the chunk shortcut and
reference-segment rules of validate.cpp:57-68,164-314 as explicit contracts."""
import numpy as np

from hyrise_amd import abi, storage
from support import oracle_validate, result_rows

# (name, row_tid, begin_cid, end_cid, visible) -- validate_visibility_test.cpp
TRUTH_TABLE = [("Impossible", 2, 2, 2, False), ("PastDelete", 42, 2, 2, False), ("Impossible2", 2, 4, 1, False),
               ("OwnDeleteUncommitted", 2, 1, 6, False), ("Impossible3", 50, 3, 1, False), ("OwnInsert", 2, 3, 3, True),
               ("PastInsertOrFutureDelete", 99, 2, 3, True), ("UncommittedInsertOrFutureInsert", 99, 3, 3, False)]


# (name, begin_cid, mutable, invalid row count, entirely visible) at snapshot commit id 1 -- validate_test.cpp:163-214: one-row chunks of
# MvccData(1, begin_cid) (tid 0, end_cid unset); a chunk that was never marked immutable has no max_begin_cid
ENTIRE_CHUNK_VISIBLE = [("ChunkNotEntirelyVisibleWithoutMaxBeginCid", 0, True, 0, False), ("ChunkNotEntirelyVisibleWithLowerSnapshotCid", 2, False, 0, False),
                        ("ChunkNotEntirelyVisibleWithInvalidRows", 0, False, 1, False), ("ChunkEntirelyVisible", 0, False, 0, True)]


def entire_chunk_case(begin, mutable, invalid):
    return storage.make_mvcc_column([0], [begin], [storage.MAX_COMMIT_ID], chunk_size=10, mutable_chunks=(0,) if mutable else (), invalid_row_counts=[invalid])


def test_is_entire_chunk_visible_of_the_reference():
    for name, begin, mutable, invalid, entirely_visible in ENTIRE_CHUNK_VISIBLE:
        got = oracle_validate(entire_chunk_case(begin, mutable, invalid), our_tid=1, snapshot_commit_id=1)
        assert (got.chunk_state[0] == abi.CHUNK_ALL_MATCH) == entirely_visible, name


def numpy_visible(tids, begins, ends, our_tid, snapshot):
    return (snapshot < ends) & ((snapshot >= begins) != (tids == our_tid))


def test_truth_table_of_the_reference():
    for name, tid, begin, end, visible in TRUTH_TABLE:
        column = storage.make_mvcc_column([tid], [begin], [end], chunk_size=10, mutable_chunks=(0,))
        got = oracle_validate(column, our_tid=2, snapshot_commit_id=2)
        assert got.total == (1 if visible else 0), name
    # the same eight rows as one chunk: positions of the visible ones
    tids, begins, ends = (np.array([row[i] for row in TRUTH_TABLE], dtype=np.uint32) for i in (1, 2, 3))
    column = storage.make_mvcc_column(tids, begins, ends, chunk_size=10, mutable_chunks=(0,))
    got = oracle_validate(column, 2, 2)
    assert [offset for _, offset in result_rows(got)] == [5, 6]


def test_chunk_shortcut_rules():
    n = 30
    tids = np.zeros(n, dtype=np.uint32)
    begins = np.full(n, 5, dtype=np.uint32)
    ends = np.full(n, storage.MAX_COMMIT_ID, dtype=np.uint32)
    ends[25] = 7                                      # chunk 2 has an invalidated row
    column = storage.make_mvcc_column(tids, begins, ends, chunk_size=10, mutable_chunks=(1,))
    got = oracle_validate(column, our_tid=9, snapshot_commit_id=6)
    # chunk 0: immutable, max_begin_cid 5 <= 6, nothing invalidated -> entirely visible; chunk 1 is mutable, chunk 2 has
    # an invalid row (still visible to snapshot 6 < end_cid 7): both are tested row by row
    assert got.chunk_state.tolist() == [abi.CHUNK_ALL_MATCH, abi.CHUNK_SCANNED, abi.CHUNK_SCANNED]
    assert got.counts.tolist() == [10, 10, 10]
    assert got.total == 20                            # the entirely visible chunk owns no RowIDs
    no_shortcut = oracle_validate(column, 9, 6, can_use_chunk_shortcut=False)   # an in-flight Delete (validate.cpp:116-127)
    assert no_shortcut.chunk_state.tolist() == [abi.CHUNK_SCANNED] * 3 and no_shortcut.total == 30
    early = oracle_validate(column, 9, 4)             # snapshot before every insert
    assert early.total == 0 and early.chunk_state.tolist() == [abi.CHUNK_SCANNED] * 3


def test_reference_segments():
    rng = np.random.default_rng(3)
    n, chunk = 2000, 300
    tids = rng.integers(0, 4, n).astype(np.uint32)
    begins = rng.integers(1, 9, n).astype(np.uint32)
    ends = np.where(rng.random(n) < 0.3, rng.integers(1, 12, n), storage.MAX_COMMIT_ID).astype(np.uint32)
    data = storage.make_mvcc_column(tids, begins, ends, chunk_size=chunk)
    rows = rng.integers(0, n, 700)
    multi = np.stack([rows // chunk, rows % chunk], axis=1).astype(np.uint32)
    single = np.stack([np.full(150, 2), rng.integers(0, chunk, 150)], axis=1).astype(np.uint32)
    reference = storage.make_reference_column(data, [multi, single, 1], [None, 2, 1])
    for our_tid, snapshot in ((2, 5), (0, 1), (3, 11)):
        got = oracle_validate(reference, our_tid, snapshot, flags=abi.SCAN_MATERIALIZE_ALL_MATCH)
        visible = numpy_visible(tids, begins, ends, our_tid, snapshot)
        expect = [np.flatnonzero(visible[multi[:, 0] * chunk + multi[:, 1]]), np.flatnonzero(visible[2 * chunk + single[:, 1]]),
                  np.flatnonzero(visible[chunk:2 * chunk])]
        got_rows = result_rows(got)
        for c in range(3):
            assert [offset for chunk, offset in got_rows if chunk == c] == expect[c].tolist(), (our_tid, snapshot, c)
