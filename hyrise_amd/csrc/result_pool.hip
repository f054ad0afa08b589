// result_pool.hip -- device memory for operator RESULTS (PosLists of scans and joins) that outlives the call that wrote it.
//
// The reference's operators return tables whose ReferenceSegments share PosLists (table_scan.cpp:207-210, join_output_writing.cpp:95-200); a
// PosList lives as long as some table references it.  Behind `_on_execute()` the adapter keeps such PosLists in HBM (DevicePosList,
// hyrise_amd/host/hyrise_host.hpp), so their memory has the lifetime of a C++ object on the host -- not of a call --
// and comes from this pool: process-wide per device, blocks handed out again in stream order (a released block carries the event of the
// releasing thread's stream; the next owner's stream waits for it), hipMalloc only when nothing fits.
//
// A join writes TWO lists at the same pair index at the same time.  Where they lie decides pk_emit's speed by up to 20 % (DESIGN.md section 4.2,
// profiles/r04_join_placement.txt, r05_placement_probe.txt): the two streams should not meet in the same memory channels -- the second list starts
// 1.25 MiB past the 2 MiB grid the first starts on -- and which stretch of HBM an allocation landed in matters by itself.  hy_result_pool_calibrate
// is the policy bench.py used to run in Python (rounds 4-5): n candidate pairs, a few joins into each, the fastest stays in the pool as the pair
// hy_result_pool_acquire_pair prefers.
#include "hy_device.hpp"

#include <algorithm>
#include <cstring>

namespace hy {
namespace {

constexpr size_t PERIOD = size_t{2} << 20;       // the grid both lists of a pair start on ...
constexpr size_t PAIR_OFFSET = size_t{5} << 18;  // ... the second one 1.25 MiB past it

struct Block {
  void* base = nullptr;         // what hipMalloc returned
  char* user = nullptr;         // what callers get
  size_t usable = 0;            // bytes from `user` on
  int device = 0;
  bool in_use = false;
  int pair = -1;                // >= 0: one list of pair `pair` (index into g_pairs)
  hipEvent_t released = nullptr;   // recorded on the releasing thread's stream
  bool has_event = false;
};
struct Pair {
  int left = -1, right = -1;    // indices into g_blocks
  uint32_t rank = 0xFFFFFFFFu;  // calibration: 0 = the fastest candidate, 1 = the median one; uncalibrated pairs come after them
  bool dead = false;
};

std::mutex g_mutex;
std::vector<Block> g_blocks;
std::vector<Pair> g_pairs;

int this_device() {
  bind_thread_device();
  int device = 0;
  (void)hipGetDevice(&device);
  return device;
}

hy_status new_block(size_t bytes, size_t offset_on_grid, int pair, int* index) {
  Block b;
  b.device = this_device();
  const size_t padded = bytes + 2 * PERIOD;
  const hipError_t err = hipMalloc(&b.base, padded);
  if (err != hipSuccess) return fail(HY_ERR_DEVICE, "result pool: hipMalloc(%zu) failed: %s", padded, hipGetErrorString(err));
  char* base = static_cast<char*>(b.base);
  b.user = base + (PERIOD - reinterpret_cast<uintptr_t>(base) % PERIOD) % PERIOD + offset_on_grid;
  b.usable = padded - static_cast<size_t>(b.user - base);
  b.pair = pair;
  b.in_use = true;
  // a slot of a freed block is used again: indices stay valid
  for (size_t i = 0; i < g_blocks.size(); ++i) {
    if (!g_blocks[i].base) { b.released = g_blocks[i].released; g_blocks[i] = b; *index = static_cast<int>(i); return HY_OK; }
  }
  g_blocks.push_back(b);
  *index = static_cast<int>(g_blocks.size() - 1);
  return HY_OK;
}

void free_block(Block& b) {
  if (b.base) (void)hipFree(b.base);
  b.base = nullptr;
  b.user = nullptr;
  b.usable = 0;
  b.in_use = false;
  b.pair = -1;
  b.has_event = false;
}

// the new owner's launches come after whatever the last owner still had queued
hy_status take(Block& b) {
  b.in_use = true;
  if (b.has_event) HY_HIP(hipStreamWaitEvent(current_stream(), b.released, 0));
  b.has_event = false;
  return HY_OK;
}

hy_status acquire_pair_locked(uint64_t rows, hy_row_id** left, hy_row_id** right, int* pair_index) {
  const size_t bytes = sizeof(hy_row_id) * static_cast<size_t>(std::max<uint64_t>(rows, 1));
  const int device = this_device();
  int best = -1;
  for (size_t p = 0; p < g_pairs.size(); ++p) {
    const Pair& pair = g_pairs[p];
    if (pair.dead) continue;
    const Block& l = g_blocks[pair.left];
    const Block& r = g_blocks[pair.right];
    if (l.in_use || r.in_use || l.device != device || l.usable < bytes || r.usable < bytes) continue;
    // calibrated pairs first (by rank); among the others the tightest fit.  A pair sized for SF10's 480 MB lists is not spent on a
    // result of a few rows unless it is all there is: small results take blocks of their own (below)
    if (l.usable > 4 * bytes + (size_t{64} << 20)) continue;
    if (best < 0 || pair.rank < g_pairs[best].rank || (pair.rank == g_pairs[best].rank && l.usable < g_blocks[g_pairs[best].left].usable)) best = static_cast<int>(p);
  }
  if (best < 0) {
    Pair pair;
    g_pairs.push_back(pair);
    best = static_cast<int>(g_pairs.size() - 1);
    int l = -1, r = -1;
    hy_status status = new_block(bytes, 0, best, &l);
    if (status == HY_OK) status = new_block(bytes, PAIR_OFFSET, best, &r);
    if (status != HY_OK) {
      if (l >= 0) free_block(g_blocks[l]);
      g_pairs[best].dead = true;
      return status;
    }
    g_pairs[best].left = l;
    g_pairs[best].right = r;
  } else {
    HY_TRY(take(g_blocks[g_pairs[best].left]));
    HY_TRY(take(g_blocks[g_pairs[best].right]));
  }
  *left = reinterpret_cast<hy_row_id*>(g_blocks[g_pairs[best].left].user);
  *right = reinterpret_cast<hy_row_id*>(g_blocks[g_pairs[best].right].user);
  if (pair_index) *pair_index = best;
  return HY_OK;
}

hy_status release_locked(void* ptr) {
  for (Block& b : g_blocks) {
    if (!b.base || b.user != ptr) continue;
    if (!b.in_use) return fail(HY_ERR_INVALID, "hy_result_pool_release: the buffer is not in use");
    if (!b.released) HY_HIP(hipEventCreateWithFlags(&b.released, hipEventDisableTiming));
    HY_HIP(hipEventRecord(b.released, current_stream()));
    b.has_event = true;
    b.in_use = false;
    return HY_OK;
  }
  return fail(HY_ERR_INVALID, "hy_result_pool_release: not a buffer of the pool");
}

void drop_pair_locked(int p) {
  Pair& pair = g_pairs[p];
  if (pair.dead) return;
  free_block(g_blocks[pair.left]);
  free_block(g_blocks[pair.right]);
  pair.dead = true;
}

}  // namespace
}  // namespace hy

using namespace hy;

extern "C" {

hy_status hy_result_pool_acquire(uint64_t bytes, void** ptr) {
  if (!ptr) return fail(HY_ERR_INVALID, "hy_result_pool_acquire: null argument");
  *ptr = nullptr;
  const size_t wanted = static_cast<size_t>(std::max<uint64_t>(bytes, 256));
  const std::lock_guard<std::mutex> lock(g_mutex);
  const int device = this_device();
  int best = -1;
  for (size_t i = 0; i < g_blocks.size(); ++i) {
    const Block& b = g_blocks[i];
    if (!b.base || b.in_use || b.pair >= 0 || b.device != device || b.usable < wanted || b.usable > 2 * wanted + (size_t{8} << 20)) continue;
    if (best < 0 || b.usable < g_blocks[best].usable) best = static_cast<int>(i);
  }
  if (best < 0) HY_TRY(new_block(wanted, 0, -1, &best));
  else HY_TRY(take(g_blocks[best]));
  *ptr = g_blocks[best].user;
  return HY_OK;
}

hy_status hy_result_pool_acquire_pair(uint64_t rows, hy_row_id** left, hy_row_id** right) {
  if (!left || !right) return fail(HY_ERR_INVALID, "hy_result_pool_acquire_pair: null argument");
  *left = *right = nullptr;
  const std::lock_guard<std::mutex> lock(g_mutex);
  return acquire_pair_locked(rows, left, right, nullptr);
}

hy_status hy_result_pool_release(void* ptr) {
  if (!ptr) return HY_OK;
  const std::lock_guard<std::mutex> lock(g_mutex);
  return release_locked(ptr);
}

hy_status hy_result_pool_trim(void) {
  const std::lock_guard<std::mutex> lock(g_mutex);
  const int device = this_device();
  HY_HIP(hipStreamSynchronize(current_stream()));
  for (size_t p = 0; p < g_pairs.size(); ++p) {
    if (g_pairs[p].dead) continue;
    const Block& l = g_blocks[g_pairs[p].left];
    const Block& r = g_blocks[g_pairs[p].right];
    if (!l.in_use && !r.in_use && l.device == device) drop_pair_locked(static_cast<int>(p));
  }
  for (Block& b : g_blocks) {
    if (b.base && !b.in_use && b.pair < 0 && b.device == device) free_block(b);
  }
  return HY_OK;
}

hy_status hy_result_pool_stats(uint64_t* held_bytes, uint64_t* in_use_bytes, uint32_t* calibrated_pairs) {
  const std::lock_guard<std::mutex> lock(g_mutex);
  uint64_t held = 0, used = 0;
  for (const Block& b : g_blocks) {
    if (!b.base) continue;
    held += b.usable;
    if (b.in_use) used += b.usable;
  }
  uint32_t calibrated = 0;
  for (const Pair& p : g_pairs) if (!p.dead && p.rank != 0xFFFFFFFFu) ++calibrated;
  if (held_bytes) *held_bytes = held;
  if (in_use_bytes) *in_use_bytes = used;
  if (calibrated_pairs) *calibrated_pairs = calibrated;
  return HY_OK;
}

hy_status hy_result_pool_calibrate(const hy_column* left, const hy_column* right, uint32_t mode, uint64_t rows, uint32_t candidates, uint32_t flags,
                                   float* ms_per_candidate, uint32_t* chosen) {
  if (!left || !right) return fail(HY_ERR_INVALID, "hy_result_pool_calibrate: null argument");
  if (candidates < 1 || candidates > 64) return fail(HY_ERR_INVALID, "hy_result_pool_calibrate: 1 .. 64 candidates");
  HY_TRY(on_this_device(left, "hy_result_pool_calibrate"));
  HY_TRY(on_this_device(right, "hy_result_pool_calibrate"));
  const bool one_list = mode == HY_JOIN_SEMI || mode == HY_JOIN_ANTI_NULL_AS_TRUE || mode == HY_JOIN_ANTI_NULL_AS_FALSE;
  const uint64_t capacity = std::max<uint64_t>(rows, 1);
  hipStream_t stream = current_stream();
  // candidates are NEW allocations (what is free in the pool already is not measured again: it keeps its rank)
  std::vector<int> pairs;
  std::vector<float> times;
  struct Buffers { hy_row_id* left; hy_row_id* right; };
  std::vector<Buffers> lists;
  {
    const std::lock_guard<std::mutex> lock(g_mutex);
    const size_t bytes = sizeof(hy_row_id) * static_cast<size_t>(capacity);
    for (uint32_t c = 0; c < candidates; ++c) {
      Pair pair;
      g_pairs.push_back(pair);
      const int p = static_cast<int>(g_pairs.size() - 1);
      int l = -1, r = -1;
      hy_status status = new_block(bytes, 0, p, &l);
      if (status == HY_OK) status = new_block(bytes, PAIR_OFFSET, p, &r);
      if (status != HY_OK) {   // out of memory: calibrate over what there is
        if (l >= 0) free_block(g_blocks[l]);
        g_pairs[p].dead = true;
        if (pairs.empty()) return status;
        break;
      }
      g_pairs[p].left = l;
      g_pairs[p].right = r;
      pairs.push_back(p);
      lists.push_back(Buffers{reinterpret_cast<hy_row_id*>(g_blocks[l].user), reinterpret_cast<hy_row_id*>(g_blocks[r].user)});
    }
  }
  DeviceBuffer slice_offsets, status_words;
  const uint32_t slice_capacity = static_cast<uint32_t>(capacity / 131070 + std::max(left->n_chunks, right->n_chunks) + 600);
  HY_TRY(slice_offsets.alloc(8 * (size_t{slice_capacity} + 2)));
  HY_TRY(status_words.alloc(sizeof(hy_join_status)));
  hipEvent_t started = nullptr, stopped = nullptr;
  HY_HIP(hipEventCreate(&started));
  HY_HIP(hipEventCreate(&stopped));
  hy_status outcome = HY_OK;
  constexpr int WARM = 3, TIMED = 4;
  for (size_t c = 0; c < pairs.size() && outcome == HY_OK; ++c) {
    hy_join_result r;
    std::memset(&r, 0, sizeof(r));
    r.mem = HY_MEM_DEVICE;
    r.left_pos = lists[c].left;
    r.right_pos = one_list ? lists[c].left : lists[c].right;
    r.capacity = capacity;
    r.slice_offsets = slice_offsets.as<uint64_t>();
    r.slice_capacity = slice_capacity;
    r.flags = HY_JOIN_ASYNC;
    r.status = status_words.as<hy_join_status>();
    for (int i = 0; i < WARM + TIMED && outcome == HY_OK; ++i) {   // (the first joins also leave the build column's key hint behind)
      if (i == WARM) (void)hipEventRecord(started, stream);
      r.radix_bits = 0xFFFFFFFFu;
      outcome = hy_join_hash(left, right, mode, &r);
    }
    (void)hipEventRecord(stopped, stream);
    if (outcome == HY_OK) outcome = hy_join_hash_finish(left, right, mode, &r);
    if (outcome != HY_OK) break;
    (void)hipEventSynchronize(stopped);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, started, stopped);
    times.push_back(ms / TIMED);
  }
  (void)hipEventDestroy(started);
  (void)hipEventDestroy(stopped);
  (void)hipStreamSynchronize(stream);
  const std::lock_guard<std::mutex> lock(g_mutex);
  if (outcome != HY_OK) {
    for (int p : pairs) drop_pair_locked(p);
    return outcome;
  }
  std::vector<size_t> order(times.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return times[a] < times[b]; });
  const size_t best = order[0], median = order[order.size() / 2];
  for (size_t c = 0; c < pairs.size(); ++c) {
    if (ms_per_candidate) ms_per_candidate[c] = times[c];
    if (c == best) {
      // earlier calibrations' winners give way: the newest measurement decides what acquire_pair prefers
      for (Pair& other : g_pairs) if (!other.dead && other.rank == 0) other.rank = 2;
      g_pairs[pairs[c]].rank = 0;
    } else if (c == median && (flags & HY_POOL_KEEP_MEDIAN)) {
      for (Pair& other : g_pairs) if (!other.dead && other.rank == 1) other.rank = 3;
      g_pairs[pairs[c]].rank = 1;
    } else {
      drop_pair_locked(pairs[c]);
      continue;
    }
    g_blocks[g_pairs[pairs[c]].left].in_use = false;
    g_blocks[g_pairs[pairs[c]].right].in_use = false;
  }
  for (size_t c = pairs.size(); c < candidates; ++c) if (ms_per_candidate) ms_per_candidate[c] = 0.f;
  if (chosen) *chosen = static_cast<uint32_t>(best);
  return HY_OK;
}

}  // extern "C"
