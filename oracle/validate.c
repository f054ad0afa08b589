/* validate.c -- CPU restatement of Hyrise's Validate operator (test infrastructure, see hy_oracle.h).
 *
 *   Validate::is_row_visible                operators/validate.cpp:47-55
 *   Validate::_is_entire_chunk_visible      validate.cpp:57-68
 *   Validate::_validate_chunks              validate.cpp:164-314
 *     data chunk:      EntireChunkPosList if the chunk is entirely visible, else the visible offsets        (:278-297)
 *     reference chunk: single-chunk pos list -> reused if the referenced chunk is entirely visible, else the
 *                      visible positions (:202-221); pos list over several chunks -> rows of entirely visible
 *                      chunks are taken without a row test, the others are tested (:222-254).  Position order.
 * The result has the layout of hyo_table_scan: (input chunk, position inside the input chunk).
 * Pinned by the reference's truth table (operators/validate_visibility_test.cpp:45-131) in tests/test_oracle_validate.py.
 */
#include <stdlib.h>
#include <string.h>

#include "hy_oracle.h"

static int row_visible(uint32_t our_tid, uint32_t snapshot, uint32_t row_tid, uint32_t begin_cid, uint32_t end_cid) {
  return snapshot < end_cid && ((snapshot >= begin_cid) != (row_tid == our_tid));
}

static int mvcc_row_visible(const hy_segment* m, uint32_t row, uint32_t our_tid, uint32_t snapshot) {
  return row_visible(our_tid, snapshot, ((const uint32_t*)m->data)[row], ((const uint32_t*)m->aux)[row], ((const uint32_t*)m->nulls)[row]);
}

static int entire_chunk_visible(const hy_segment* m, uint32_t snapshot, uint32_t can_use_chunk_shortcut) {
  if (!can_use_chunk_shortcut) return 0;
  const int is_mutable = (m->ref_chunk_id >> 31) != 0;
  const uint32_t invalid_rows = m->ref_chunk_id & 0x7FFFFFFFu, max_begin_cid = m->aux_size;
  return !is_mutable && snapshot >= max_begin_cid && invalid_rows == 0;
}

int32_t hyo_validate(const hyo_column* column, uint32_t our_tid, uint32_t snapshot, uint32_t can_use_chunk_shortcut, hy_scan_result* result) {
  uint64_t cursor = 0;
  for (uint32_t c = 0; c < column->n_chunks; ++c) {
    const hy_segment* seg = &column->segments[c];
    uint8_t state = HY_CHUNK_SCANNED;
    uint32_t count = 0;
    result->offsets[c] = cursor;
    if (cursor + seg->size > result->capacity) return HY_ERR_CAPACITY;
    hy_row_id* out = result->matches + cursor;
    if (seg->encoding == HY_ENC_MVCC) {
      if (entire_chunk_visible(seg, snapshot, can_use_chunk_shortcut)) {
        state = HY_CHUNK_ALL_MATCH;
        count = seg->size;
        if (result->flags & HY_SCAN_MATERIALIZE_ALL_MATCH)
          for (uint32_t i = 0; i < seg->size; ++i) { out[i].chunk_id = c; out[i].chunk_offset = i; }
      } else {
        for (uint32_t i = 0; i < seg->size; ++i)
          if (mvcc_row_visible(seg, i, our_tid, snapshot)) { out[count].chunk_id = c; out[count].chunk_offset = i; ++count; }
      }
    } else if (seg->encoding == HY_ENC_REFERENCE) {
      const hyo_column* referenced = (const hyo_column*)seg->ref;
      const hy_row_id* pos_list = (const hy_row_id*)seg->data;
      if (seg->ref_chunk_id != 0xFFFFFFFFu && seg->size > 0) { /* single referenced chunk (:202-221) */
        const hy_segment* m = &referenced->segments[seg->ref_chunk_id];
        if (entire_chunk_visible(m, snapshot, can_use_chunk_shortcut)) {
          state = HY_CHUNK_ALL_MATCH;
          count = seg->size;
          if (result->flags & HY_SCAN_MATERIALIZE_ALL_MATCH)
            for (uint32_t i = 0; i < seg->size; ++i) { out[i].chunk_id = c; out[i].chunk_offset = i; }
        } else {
          for (uint32_t i = 0; i < seg->size; ++i) {
            const uint32_t row = pos_list ? pos_list[i].chunk_offset : i;
            if (row != 0xFFFFFFFFu && mvcc_row_visible(m, row, our_tid, snapshot)) { out[count].chunk_id = c; out[count].chunk_offset = i; ++count; }
          }
        }
      } else { /* several referenced chunks (:222-254) */
        for (uint32_t i = 0; i < seg->size; ++i) {
          const hy_row_id r = pos_list[i];
          if (r.chunk_offset == 0xFFFFFFFFu) continue; /* NULL_ROW_ID: not a row of the referenced table */
          const hy_segment* m = &referenced->segments[r.chunk_id];
          if (entire_chunk_visible(m, snapshot, can_use_chunk_shortcut) || mvcc_row_visible(m, r.chunk_offset, our_tid, snapshot)) {
            out[count].chunk_id = c; out[count].chunk_offset = i; ++count;
          }
        }
      }
    } else {
      return HY_ERR_INVALID;
    }
    result->counts[c] = count;
    result->chunk_state[c] = state;
    if (!(state == HY_CHUNK_ALL_MATCH && !(result->flags & HY_SCAN_MATERIALIZE_ALL_MATCH))) cursor += count;
  }
  result->offsets[column->n_chunks] = cursor;
  result->total_matches = cursor;
  return HY_OK;
}
