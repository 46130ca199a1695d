// Micro-benchmark: cycles per wave-wide ds_read_b128 / ds_read_b64 / ds_read_b32 for several lane -> address maps.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_b128 scripts/micro/lds_b128.hip && /tmp/lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int BYTES>
__global__ __launch_bounds__(256) void k(const int* __restrict__ ofs, unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  const int o = ofs[threadIdx.x & 63];
  unsigned acc = 0;
  const unsigned base = (unsigned)(size_t)smem;   // LDS address (low 32 bits of the generic pointer are the offset)
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned a = base + ((o + u * 2048) & 32767);
      if constexpr (BYTES == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
      else if constexpr (BYTES == 8) { u32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a)); v[u] = (u32x4){t.x, t.y, 0u, 0u}; }
      else { unsigned t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(a)); v[u] = (u32x4){t, 0u, 0u, 0u}; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x;
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  struct Pat { const char* name; int bytes; int (*f)(int); };
  Pat pats[] = {
      {"b128 linear lane*16", 16, [](int l) { return l * 16; }},
      {"b128 stride 32B (jn*32, kq*16)", 16, [](int l) { return (l & 15) * 32 + (l >> 4) * 16; }},
      {"b128 conv0 swizzled", 16, [](int l) { int x = 2 * (l & 15) + (l >> 4); x ^= (x >> 3) & 1; return x * 16; }},
      {"b128 lane&15 *16 (4x broadcast)", 16, [](int l) { return (l & 15) * 16; }},
      {"b128 lane&15 *16 + kq*256", 16, [](int l) { return (l & 15) * 16 + (l >> 4) * 256; }},
      {"b128 stride 64B", 16, [](int l) { return l * 64; }},
      {"b128 stride 64B xor", 16, [](int l) { return l * 64 + (((l >> 1) & 3) * 16); }},
      {"b128 gemm (jn*64 + slot swz)", 16, [](int l) { int jn = l & 15, kq = l >> 4; return jn * 64 + ((kq ^ (((jn >> 3) & 1) * 3)) * 16); }},
      {"b128 gemm plain (jn*64 + kq*16)", 16, [](int l) { int jn = l & 15, kq = l >> 4; return jn * 64 + kq * 16; }},
      {"b128 stride 128B (worst)", 16, [](int l) { return l * 128; }},
      {"b64 linear", 8, [](int l) { return l * 8; }},
      {"b64 stride 16B", 8, [](int l) { return l * 16; }},
      {"b64 prob (y*128 + xg*16 + 16)", 8, [](int l) { return (l >> 3) * 128 + (l & 7) * 16 + 16; }},
      {"b128 prob (y*128 + xg*16)", 16, [](int l) { return (l >> 3) * 128 + (l & 7) * 16; }},
      {"b32 linear", 4, [](int l) { return l * 4; }},
      {"b32 stride 8B", 4, [](int l) { return l * 8; }},
      {"b32 same bank (stride 128B)", 4, [](int l) { return l * 128; }},
      {"b32 f32 mfma B (jn*136 + kq*4)", 4, [](int l) { return (l & 15) * 136 + (l >> 4) * 4; }},
  };
  int* d_ofs; unsigned* d_out; long long* d_cyc;
  hipMalloc(&d_ofs, 64 * 4); hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_cyc, 256 * 8);
  const int iters = 2000;
  for (auto& p : pats) {
    std::vector<int> o(64);
    for (int l = 0; l < 64; ++l) o[l] = p.f(l);
    hipMemcpy(d_ofs, o.data(), 256, hipMemcpyHostToDevice);
    for (int blocks_per_cu = 1; blocks_per_cu <= 1; ++blocks_per_cu) {
      // one block (4 waves, one per SIMD) on a CU: all four waves compete for the CU's LDS
      for (int rep = 0; rep < 2; ++rep) {
        if (p.bytes == 16) k<16><<<1, 256>>>(d_ofs, d_out, d_cyc, iters);
        else if (p.bytes == 8) k<8><<<1, 256>>>(d_ofs, d_out, d_cyc, iters);
        else k<4><<<1, 256>>>(d_ofs, d_out, d_cyc, iters);
        hipDeviceSynchronize();
      }
      long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
      // clock64 = s_memtime at 100 MHz constant?  report raw ticks per (4 waves x 8 x iters) reads
      printf("%-40s %8.3f ticks per wave-read (x4 waves sharing the LDS)\n", p.name, (double)c / (8.0 * iters * 4));
    }
  }
  return 0;
}
