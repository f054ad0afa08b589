// Attainable READ bandwidth probe for pass 1 of the PK-FK join (pk_count) on one MI355X: 120 MB of 2-byte words read once, in the launch
// shapes the kernel could take (not part of the product; `hipcc --offload-arch=gfx950 -O3 tools/hbm_read.hip -o tools/hbm_read_bin`).
//   tile      one workgroup of THREADS threads per TILE bytes, 16-byte loads, all of a lane's loads in flight at once
//   xcd       the same, workgroup b reads tile (b % 8) * (tiles / 8) + b / 8 (pk_count's order: an eighth of the column per XCD)
//   stream    persistent workgroups (GRID of them), grid-stride over 16 KB pieces
// Four copies of the column in rotation (480 MB > the 256 MiB memory-side cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS, bool XCD>
__global__ void tile_read(const u32x4* __restrict__ in, unsigned n_tiles, unsigned* sink) {
  const unsigned tile = XCD ? (blockIdx.x & 7) * (n_tiles / 8) + (blockIdx.x >> 3) : blockIdx.x;
  const u32x4* src = in + static_cast<size_t>(tile) * LOADS * blockDim.x;
  u32x4 v[LOADS];
#pragma unroll
  for (int i = 0; i < LOADS; ++i) v[i] = src[i * blockDim.x + threadIdx.x];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ in, size_t vectors, unsigned* sink) {
  unsigned acc = 0;
  for (size_t base = static_cast<size_t>(blockIdx.x) * 1024; base < vectors; base += static_cast<size_t>(gridDim.x) * 1024) {
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = in[base + i * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  const size_t bytes = 120ull << 20, vectors = bytes / 16;
  const int copies = 4;
  u32x4* in;
  unsigned* sink;
  CHECK(hipMalloc(&in, bytes * copies));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(in, 1, bytes * copies));
  hipEvent_t start, stop;
  CHECK(hipEventCreate(&start));
  CHECK(hipEventCreate(&stop));
  auto timed = [&](const char* name, auto launch) {
    for (int i = 0; i < 4; ++i) launch(in + (i % copies) * vectors);
    float total = 0;
    const int reps = 16;
    for (int i = 0; i < reps; ++i) {
      CHECK(hipEventRecord(start));
      launch(in + (i % copies) * vectors);
      CHECK(hipEventRecord(stop));
      CHECK(hipEventSynchronize(stop));
      float ms;
      CHECK(hipEventElapsedTime(&ms, start, stop));
      total += ms;
    }
    printf("%-64s %7.1f us  %6.0f GB/s\n", name, total / reps * 1e3, bytes / (total / reps * 1e-3) / 1e9);
  };
  timed("tile: 16 KB per workgroup of 256 threads, 4 loads per lane", [&](const u32x4* p) { hipLaunchKernelGGL((tile_read<4, false>), dim3(vectors / 1024), dim3(256), 0, 0, p, unsigned(vectors / 1024), sink); });
  timed("xcd:  the same, an eighth of the column per XCD", [&](const u32x4* p) { hipLaunchKernelGGL((tile_read<4, true>), dim3(vectors / 1024), dim3(256), 0, 0, p, unsigned(vectors / 1024), sink); });
  timed("tile: 32 KB per workgroup of 256 threads, 8 loads per lane", [&](const u32x4* p) { hipLaunchKernelGGL((tile_read<8, false>), dim3(vectors / 2048), dim3(256), 0, 0, p, unsigned(vectors / 2048), sink); });
  timed("xcd:  the same, an eighth of the column per XCD", [&](const u32x4* p) { hipLaunchKernelGGL((tile_read<8, true>), dim3(vectors / 2048), dim3(256), 0, 0, p, unsigned(vectors / 2048), sink); });
  timed("tile: 16 KB per workgroup of 512 threads, 2 loads per lane", [&](const u32x4* p) { hipLaunchKernelGGL((tile_read<2, false>), dim3(vectors / 1024), dim3(512), 0, 0, p, unsigned(vectors / 1024), sink); });
  timed("tile: 64 KB per workgroup of 512 threads, 8 loads per lane", [&](const u32x4* p) { hipLaunchKernelGGL((tile_read<8, false>), dim3(vectors / 4096), dim3(512), 0, 0, p, unsigned(vectors / 4096), sink); });
  for (unsigned grid : {256u, 512u, 1024u, 2048u, 4096u}) {
    char name[96];
    snprintf(name, sizeof(name), "stream: %u persistent workgroups of 256 threads", grid);
    timed(name, [&](const u32x4* p) { hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, 0, p, vectors, sink); });
  }
  return 0;
}
