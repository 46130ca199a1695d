// Vector-memory issue cost of partially masked 16-byte gathers on gfx950 (developer micro-benchmark; hipcc --offload-arch=gfx950).
// Question behind it (warp kernel): a footprint reload is four global_load_dwordx4 with 8 lanes per pixel; when
// only some of a wave's 8 pixels changed their footprint the other pixels' lanes are masked.  Is such a load charged per wave
// instruction (16 cycles of the CU's 64 B/clk L1 path) or per active lane group?
//
// One 256/512/1024-thread workgroup per CU; every wave issues groups of 8 independent loads from a 16 KB window of its own
// workgroup (L1-resident after the first pass) -- lane = (pixel = lane >> 3, 16-byte part = lane & 7), each pixel reads one
// 128-byte cell picked by a per-load pseudo-random index, as the warp kernel does.  Prints cycles per wave load instruction
// per CU for a set of exec masks.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int WIDTH>   // bytes per lane: 16 / 8 / 4
__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, long long* out, float* sink, int iters,
                                           unsigned long long mask, int window_cells) {
  const int lane = threadIdx.x & 63;
  const bool active = (mask >> lane) & 1ull;
  const char* base = buf + (size_t)blockIdx.x * window_cells * 128;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  // per (pixel, load slot) cell walk: o <- (o + odd number of cells) mod window, 3 VALU instructions per load
  unsigned o[8], inc[8];
  const unsigned wmask = (unsigned)window_cells * 128u - 1u;      // window_cells is a power of two
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned h = ((threadIdx.x >> 3) * 8u + j) * 2654435761u + 12345u;
    o[j] = ((h >> 9) * 128u) & wmask;
    inc[j] = (((h >> 20) | 1u) * 128u) & wmask;
  }
  long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    unsigned ofs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = (o[j] + inc[j]) & wmask;
      ofs[j] = o[j] + (lane & 7) * 16u;
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (WIDTH == 16) acc += *reinterpret_cast<const f4*>(base + ofs[j]);
        if constexpr (WIDTH == 8) { const float2 v = *reinterpret_cast<const float2*>(base + ofs[j]); acc[0] += v.x; acc[1] += v.y; }
        if constexpr (WIDTH == 4) acc[0] += *reinterpret_cast<const float*>(base + ofs[j]);
      }
    }
  }
  long long c1 = __builtin_readcyclecounter();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int WIDTH>
static void run(const char* name, unsigned long long mask, const char* buf, long long* d, float* s, int n_cu, int window_cells) {
  const int iters = 4000;
  printf("%-34s w=%2d B ", name, WIDTH);
  for (int waves : {4, 8, 16}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<WIDTH><<<n_cu, waves * 64, 0, 0>>>(buf, d, s, 200, mask, window_cells);   // warm
    hipEventRecord(e0, 0);
    k<WIDTH><<<n_cu, waves * 64, 0, 0>>>(buf, d, s, iters, mask, window_cells);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(16 * n_cu);
    hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < n_cu; ++b) for (int w = 0; w < waves; ++w) cyc += (double)h[b * 16 + w];
    cyc /= (double)n_cu * waves;                       // cycles per wave for its iters * 8 loads
    const double per_cu = cyc / ((double)iters * 8 * waves);   // cycles per wave load instruction per CU (waves run concurrently)
    printf(" | %2d waves/CU: %6.2f cyc/load/CU (%.3f ms)", waves, per_cu, ms);
  }
  printf("\n");
}

int main() {
  int n_cu = 256;
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
  const int window_cells = 128;                           // 16 KB per workgroup: L1-resident
  char* buf; long long* d; float* s;
  hipMalloc(&buf, (size_t)n_cu * window_cells * 128 + 4096);
  hipMemset(buf, 0, (size_t)n_cu * window_cells * 128 + 4096);
  hipMalloc(&d, sizeof(long long) * 16 * n_cu);
  hipMalloc(&s, sizeof(float) * 1024 * n_cu);
  struct { const char* name; unsigned long long m; } pats[] = {
      {"all 8 pixels", ~0ull},
      {"pixels 0-3 (lanes 0-31)", 0x00000000ffffffffull},
      {"pixels 0,2,4,6", 0x00ff00ff00ff00ffull},
      {"pixels 0,1", 0x000000000000ffffull},
      {"pixel 0", 0x00000000000000ffull},
      {"pixel 5", 0x0000ff0000000000ull},
      {"lanes 0-3 of every pixel", 0x0f0f0f0f0f0f0f0full},
      {"lane 0 of every pixel", 0x0101010101010101ull},
  };
  for (auto& p : pats) run<16>(p.name, p.m, buf, d, s, n_cu, window_cells);
  for (auto& p : pats) run<8>(p.name, p.m, buf, d, s, n_cu, window_cells);
  run<4>("all 8 pixels", ~0ull, buf, d, s, n_cu, window_cells);
  // the same gathers from a window far beyond L1 (4 MB per workgroup would not fit: use all of L2 instead) are not measured here
  return 0;
}
