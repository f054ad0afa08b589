// Does the time of pk_emit's write pattern depend on WHICH allocation it writes into?  (not part of the product;
// `hipcc --offload-arch=gfx950 -O3 tools/write_fronts.hip -o tools/write_fronts_bin && tools/write_fronts_bin`)
// The pattern: tiles of 32 partitions x one 2 KiB run per output list (two lists, the second 1.25 MiB past a 2 MiB boundary), partition p's
// runs of consecutive tiles adjacent -- 0.96 GB per launch.  Tile order: one front per XCD (block b -> the (b % 8)-th eighth of the tiles), or
// one front for the device (block b -> tile b).  Arenas: six hipMalloc'ed one after the other; six pieces of one allocation; six allocated
// after the pool of free device memory was cut up by freeing every other one of 400 32 MiB blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MAP>
__global__ __launch_bounds__(512) void runs(u32x4* left, u32x4* right, unsigned n_tiles, unsigned run_vectors, unsigned partitions) {
  const unsigned tile = MAP == 0 ? (blockIdx.x % 8) * (n_tiles / 8) + blockIdx.x / 8 : blockIdx.x;
  const unsigned per_tile = partitions * run_vectors;
  for (unsigned i = threadIdx.x; i < per_tile; i += 512) {
    const unsigned partition = i / run_vectors, within = i % run_vectors;
    const size_t position = (static_cast<size_t>(partition) * n_tiles + tile) * run_vectors + within;
    __builtin_nontemporal_store(u32x4{tile, i, 1, 2}, left + position);
    __builtin_nontemporal_store(u32x4{tile, i, 3, 4}, right + position);
  }
}

static hipEvent_t start, stop;
static const size_t LIST = 480ull << 20, ARENA = 2 * LIST + (6ull << 20);

template <int MAP>
static float measure(char* arena) {
  const unsigned run_vectors = 2048 / 16, partitions = 32;
  const unsigned n_tiles = static_cast<unsigned>(LIST / (partitions * 2048)) / 8 * 8;
  char* first = arena + (-reinterpret_cast<uintptr_t>(arena) % (2ull << 20));
  char* second = first + (LIST + (2ull << 20) - 1) / (2ull << 20) * (2ull << 20) + (5ull << 18);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(runs<MAP>, dim3(n_tiles), dim3(512), 0, 0, reinterpret_cast<u32x4*>(first), reinterpret_cast<u32x4*>(second), n_tiles, run_vectors, partitions);
  float total = 0;
  const int reps = 8;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(start));
    hipLaunchKernelGGL(runs<MAP>, dim3(n_tiles), dim3(512), 0, 0, reinterpret_cast<u32x4*>(first), reinterpret_cast<u32x4*>(second), n_tiles, run_vectors, partitions);
    CHECK(hipEventRecord(stop));
    CHECK(hipEventSynchronize(stop));
    float ms;
    CHECK(hipEventElapsedTime(&ms, start, stop));
    total += ms;
  }
  return total / reps * 1e3f;
}

static void report(const char* what, const std::vector<char*>& arenas) {
  printf("%-44s per-XCD fronts:", what);
  for (char* a : arenas) printf(" %6.1f", measure<0>(a));
  printf("   one front:");
  for (char* a : arenas) printf(" %6.1f", measure<1>(a));
  printf("  us\n");
  fflush(stdout);
}

int main() {
  CHECK(hipEventCreate(&start));
  CHECK(hipEventCreate(&stop));
  const int K = 6;
  char* inputs = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void**>(&inputs), 1500ull << 20));   // (what a process holds before it allocates results)
  {
    std::vector<char*> arenas(K);
    for (auto& a : arenas) CHECK(hipMalloc(reinterpret_cast<void**>(&a), ARENA));
    report("six allocations", arenas);
    report("again", arenas);
    for (auto a : arenas) CHECK(hipFree(a));
  }
  {
    char* big = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&big), K * ARENA + (1ull << 30)));
    std::vector<char*> arenas(K);
    for (int i = 0; i < K; ++i) arenas[i] = big + i * ARENA;
    report("six pieces of one allocation", arenas);
    char* aligned = big + (-reinterpret_cast<uintptr_t>(big) % (1ull << 30));
    for (int i = 0; i < K; ++i) arenas[i] = aligned + i * ARENA;
    report("... from a 1 GiB boundary", arenas);
    CHECK(hipFree(big));
  }
  {
    std::vector<char*> blocks(400);
    for (auto& b : blocks) CHECK(hipMalloc(reinterpret_cast<void**>(&b), 32ull << 20));
    for (size_t i = 0; i < blocks.size(); i += 2) CHECK(hipFree(blocks[i]));
    std::vector<char*> arenas(K);
    for (auto& a : arenas) CHECK(hipMalloc(reinterpret_cast<void**>(&a), ARENA));
    report("six allocations, free memory cut up", arenas);
    for (auto a : arenas) CHECK(hipFree(a));
    for (size_t i = 1; i < blocks.size(); i += 2) CHECK(hipFree(blocks[i]));
  }
  return 0;
}
