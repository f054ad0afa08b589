// Attainable WRITE bandwidth probe for the join's output pattern on one MI355X (not part of the product;
// `hipcc --offload-arch=gfx950 -O3 tools/hbm_write.hip -o /tmp/hbm_write && /tmp/hbm_write`).
//   stream   every workgroup writes one contiguous 64 KiB piece with 16-byte nontemporal stores
//   runs     every workgroup ("tile") writes 2 arrays x 128 partitions x one RUN-byte run: partition p's runs of consecutive tiles
//            are adjacent (the layout rt_probe_emit produces: 4096 pairs per tile, 128 partitions, 256-byte runs per array)
// 0.96 GB per launch into buffers rotated over three allocations (nothing is found in the Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stream(u32x4* out) {
  u32x4* dst = out + static_cast<size_t>(blockIdx.x) * 4096;   // 64 KiB per workgroup
  for (unsigned i = threadIdx.x; i < 4096; i += 512) __builtin_nontemporal_store(u32x4{blockIdx.x, i, 1, 2}, dst + i);
}

// run_vectors: 16-byte vectors per run (16 = 256 bytes); a tile writes 2 x 128 runs = 2 x 128 x run_vectors vectors
template <int STORE>   // 0: nontemporal, 1: plain (write-back L2), 2: plain for the vectors of a run's first and last line, nontemporal inside
__global__ __launch_bounds__(512) void runs(u32x4* left, u32x4* right, unsigned n_tiles, unsigned run_vectors, unsigned partitions = 128) {
  const unsigned tile = (blockIdx.x % 8) * (n_tiles / 8) + blockIdx.x / 8;          // XCD x works on the x-th eighth of the tiles
  const unsigned per_tile = partitions * run_vectors;
  for (unsigned i = threadIdx.x; i < per_tile; i += 512) {
    const unsigned partition = i / run_vectors, within = i % run_vectors;
    const size_t position = (static_cast<size_t>(partition) * n_tiles + tile) * run_vectors + within;
    const bool edge = STORE == 1 || (STORE == 2 && (within < 8 || within + 8 >= run_vectors));
    if (edge) { left[position] = u32x4{tile, i, 1, 2}; right[position] = u32x4{tile, i, 3, 4}; }
    else { __builtin_nontemporal_store(u32x4{tile, i, 1, 2}, left + position); __builtin_nontemporal_store(u32x4{tile, i, 3, 4}, right + position); }
  }
}

int main() {
  const size_t bytes = 960ull << 20;
  u32x4* buffers[3];
  for (auto& b : buffers) CHECK(hipMalloc(reinterpret_cast<void**>(&b), bytes));
  hipEvent_t start, stop;
  CHECK(hipEventCreate(&start));
  CHECK(hipEventCreate(&stop));
  auto timed = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch(buffers[i % 3]);
    float total = 0;
    const int reps = 12;
    for (int i = 0; i < reps; ++i) {
      CHECK(hipEventRecord(start));
      launch(buffers[i % 3]);
      CHECK(hipEventRecord(stop));
      CHECK(hipEventSynchronize(stop));
      float ms;
      CHECK(hipEventElapsedTime(&ms, start, stop));
      total += ms;
    }
    printf("%-34s %7.1f us  %6.0f GB/s\n", name, total / reps * 1e3, bytes / (total / reps * 1e-3) / 1e9);
  };
  timed("stream, 64 KiB per workgroup", [&](u32x4* b) { hipLaunchKernelGGL(stream, dim3(bytes / 65536), dim3(512), 0, 0, b); });
  for (unsigned run_bytes : {128u, 256u, 512u, 1024u, 4096u}) {
    const unsigned run_vectors = run_bytes / 16;
    const unsigned n_tiles = static_cast<unsigned>(bytes / 2 / (128 * run_bytes)) / 8 * 8;
    char name[64];
    snprintf(name, sizeof(name), "runs of %u bytes, %u tiles", run_bytes, n_tiles);
    timed(name, [&](u32x4* b) { hipLaunchKernelGGL(runs<0>, dim3(n_tiles), dim3(512), 0, 0, b, b + bytes / 32, n_tiles, run_vectors); });
    // the same runs 48 bytes off their lines: every run starts and ends with a partial 128-byte line that a NEIGHBOURING tile completes
    snprintf(name, sizeof(name), "  ... 48 bytes off the line grid");
    timed(name, [&](u32x4* b) { hipLaunchKernelGGL(runs<0>, dim3(n_tiles), dim3(512), 0, 0, b + 3, b + bytes / 32 + 3, n_tiles - 8, run_vectors); });
    snprintf(name, sizeof(name), "  ... off the grid, plain stores");
    timed(name, [&](u32x4* b) { hipLaunchKernelGGL(runs<1>, dim3(n_tiles), dim3(512), 0, 0, b + 3, b + bytes / 32 + 3, n_tiles - 8, run_vectors); });
    snprintf(name, sizeof(name), "  ... off the grid, plain edge lines");
    timed(name, [&](u32x4* b) { hipLaunchKernelGGL(runs<2>, dim3(n_tiles), dim3(512), 0, 0, b + 3, b + bytes / 32 + 3, n_tiles - 8, run_vectors); });
  }
  // the shape of pk_emit at config 3: 32 populated partitions, 8192-row tiles -> 2 KB runs per array (and 4096 / 16384-row tiles)
  for (unsigned run_bytes : {1024u, 2048u, 4096u}) {
    const unsigned run_vectors = run_bytes / 16, partitions = 32;
    const unsigned n_tiles = static_cast<unsigned>(bytes / 2 / (partitions * run_bytes)) / 8 * 8;
    char name[96];
    snprintf(name, sizeof(name), "32 partitions, runs of %u bytes, %u tiles", run_bytes, n_tiles);
    timed(name, [&](u32x4* b) { hipLaunchKernelGGL(runs<0>, dim3(n_tiles), dim3(512), 0, 0, b, b + bytes / 32, n_tiles, run_vectors, partitions); });
    timed("  ... 48 bytes off the line grid", [&](u32x4* b) { hipLaunchKernelGGL(runs<0>, dim3(n_tiles), dim3(512), 0, 0, b + 3, b + bytes / 32 + 3, n_tiles - 8, run_vectors, partitions); });
    timed("  ... off the grid, plain stores", [&](u32x4* b) { hipLaunchKernelGGL(runs<1>, dim3(n_tiles), dim3(512), 0, 0, b + 3, b + bytes / 32 + 3, n_tiles - 8, run_vectors, partitions); });
    timed("  ... off the grid, plain edge lines", [&](u32x4* b) { hipLaunchKernelGGL(runs<2>, dim3(n_tiles), dim3(512), 0, 0, b + 3, b + bytes / 32 + 3, n_tiles - 8, run_vectors, partitions); });
  }
  return 0;
}
