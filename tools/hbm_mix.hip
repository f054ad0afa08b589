// Attainable-bandwidth probe for the scan's traffic mix on one MI355X (not part of the product; `hipcc --offload-arch=gfx950
// -O3 tools/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix`).  Streams R bytes in and Wr bytes out per workgroup with the
// same shape as scan_slices (256 threads, 16 B loads, 8 B stores, one contiguous region per workgroup) but no ALU work,
// no LDS and no barriers: what the memory system gives for  read : write = 2 B : 8 B * selectivity.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Each workgroup: rows_per_wg rows of 2 B read, out_per_wg RowIDs of 8 B written.
template <int MODE>   // bit 0: 16-byte stores, bit 1: nontemporal stores, bit 2: nontemporal loads
__global__ __launch_bounds__(256) void mix(const u32x4* __restrict__ in, uint2* __restrict__ out, unsigned rows_per_wg, unsigned out_per_wg,
                                            unsigned n_parts, unsigned* __restrict__ sink, unsigned region_stride) {
  unsigned acc = 0;
  for (unsigned part = blockIdx.x; part < n_parts; part += gridDim.x) {
    const u32x4* src = in + static_cast<size_t>(part) * (rows_per_wg / 8);
    uint2* dst = out + static_cast<size_t>(part) * region_stride;   // chunk regions, like the scan
    const unsigned loads = rows_per_wg / 8, stores = out_per_wg;
    const unsigned steps = 8;
    for (unsigned s = 0; s < steps; ++s) {
      for (unsigned i = s * (loads / steps) + threadIdx.x; i < (s + 1) * (loads / steps); i += 256) {
        const u32x4 v = (MODE & 4) ? __builtin_nontemporal_load(src + i) : src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
      if constexpr (MODE & 1) {
        u32x4* dst4 = reinterpret_cast<u32x4*>(dst);
        for (unsigned i = s * (stores / steps) / 2 + threadIdx.x; i < (s + 1) * (stores / steps) / 2; i += 256) {
          u32x4 v = {part, 2 * i + (acc & 1), part, 2 * i + 1};
          if constexpr (MODE & 2) __builtin_nontemporal_store(v, dst4 + i); else dst4[i] = v;
        }
      } else {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2* dst2 = reinterpret_cast<u32x2*>(dst);
        for (unsigned i = s * (stores / steps) + threadIdx.x; i < (s + 1) * (stores / steps); i += 256) {
          u32x2 v = {part, i + (acc & 1)};
          if constexpr (MODE & 2) __builtin_nontemporal_store(v, dst2 + i); else dst2[i] = v;
        }
      }
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
  // argv[1] = number of input copies scanned in rotation (default 3: every launch reads its input from HBM, like bench.py's rotating
  // column copies; 1: the 120 MB input stays in the 256 MiB memory-side cache)
  const int copies = argc > 1 ? atoi(argv[1]) : 3;
  const unsigned total_rows = 916u * 65536u;
  u32x4* in;
  uint2* out;
  unsigned* sink;
  CHECK(hipMalloc(&in, size_t{total_rows} * 2 * copies));
  CHECK(hipMalloc(&out, (size_t{total_rows} + 4096u * 3665u) * 8));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(in, 1, size_t{total_rows} * 2 * copies));
  const size_t copy_words = size_t{total_rows} * 2 / 16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  typedef void (*Kernel)(const u32x4*, uint2*, unsigned, unsigned, unsigned, unsigned*, unsigned);
  const Kernel kernels[8] = {mix<0>, mix<1>, mix<2>, mix<3>, mix<4>, mix<5>, mix<6>, mix<7>};
  const char* mode_names[8] = {"8 B write-back stores", "16 B write-back stores", "8 B nontemporal stores", "16 B nontemporal stores", "8 B write-back stores, nontemporal loads",
                               "16 B write-back stores, nontemporal loads", "8 B nontemporal stores, nontemporal loads", "16 B nontemporal stores, nontemporal loads"};
  // the scan's traffic (2 B read per row, 8 B written per match) for three selectivities, workgroups that own 65536 / 32768 / 16384 rows
  // (one, two, four per Hyrise chunk), every store flavour
  for (double sel : {0.15, 0.43, 0.986}) {
    for (unsigned rows : {65536u, 32768u, 16384u}) {
      const unsigned n_parts = total_rows / rows, stride = rows + 1040u;
      for (unsigned mode : {3u, 1u, 2u, 7u}) {
        const Kernel mix = kernels[mode];
        const unsigned out_per_wg = static_cast<unsigned>(rows * sel) / 8 * 8;
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(mix, dim3(n_parts), dim3(256), 0, 0, in + (i % copies) * copy_words, out, rows, out_per_wg, n_parts, sink, stride);
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        const int reps = 16;
        for (int i = 0; i < reps; ++i) {
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(mix, dim3(n_parts), dim3(256), 0, 0, in + (i % copies) * copy_words, out, rows, out_per_wg, n_parts, sink, stride);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
          sum += ms;
        }
        const double bytes = double(n_parts) * (rows * 2.0 + out_per_wg * 8.0);
        printf("selectivity %.3f  %5u workgroups x %5u rows  %-44s %6.1f MB  avg %6.1f us (%4.0f GB/s)  best %6.1f us (%4.0f GB/s)\n", sel, n_parts, rows, mode_names[mode], bytes / 1e6,
               sum / reps * 1e3, bytes / (sum / reps * 1e-3) / 1e9, best * 1e3, bytes / (best * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
