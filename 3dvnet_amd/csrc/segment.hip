// Row B4's pooling (scenemodeling.py:129-141: scatter(x, idx, reduce='max') over the points of a voxel) without atomics.
// The gather-GEMM's fused scatter-max issues one atomic per (point, channel): 25.7 M of them per PointNet layer at cfg3, 0.43 of
// the layer's 0.53 ms, and twice that when the points of a voxel sit in consecutive rows (same address from neighbouring
// lanes).  Instead the points are grouped by voxel ONCE per forward (stable radix sort of (voxel id, row) -> row list + offsets,
// `v3d_segment_csr`), every layer stores its [M, N] output (it does anyway: the next layer reads it) and `v3d_segment_max_f32`
// reduces each voxel's rows: 32 lanes x float4 per voxel, whole 512-byte rows gathered through the row list, no contention.
// max is order independent: the pooled features equal the atomic version's bit for bit.
#include "v3d_common.h"      // <cstring> before rocprim: its texture iterator calls the host memset

#include <rocprim/rocprim.hpp>

namespace {

__global__ __launch_bounds__(256) void iota_kernel(unsigned* v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = (unsigned)i;
}

// offsets[s] = first position of a key >= s in the sorted key list (binary search, one thread per segment; empty segments get
// an empty range); offsets[n_seg] = n
__global__ __launch_bounds__(256) void segment_offsets_kernel(const unsigned* __restrict__ sorted, int n, int n_seg,
                                                              int* __restrict__ offsets) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s > n_seg) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sorted[mid] < (unsigned)s) lo = mid + 1; else hi = mid;
  }
  offsets[s] = lo;
}

// LPS lanes (x float4) per segment, 256 / LPS segments per workgroup
template <int LPS>
__global__ __launch_bounds__(256) void segment_max_kernel(const float* __restrict__ src, int ld, const int* __restrict__ perm,
                                                          const int* __restrict__ offsets, int n_seg, int N,
                                                          float* __restrict__ out, int ld_out) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int seg = blockIdx.x * (256 / LPS) + threadIdx.x / LPS;
  const int c = (threadIdx.x % LPS) * 4;
  if (seg >= n_seg || c >= N) return;
  const int r0 = offsets[seg], r1 = offsets[seg + 1];
  f32x4 m0 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, m1 = m0;
  int r = r0;
  for (; r + 1 < r1; r += 2) {            // two independent row gathers in flight
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)perm[r] * ld + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + (size_t)perm[r + 1] * ld + c);
    m0 = __builtin_elementwise_max(m0, a);
    m1 = __builtin_elementwise_max(m1, b);
  }
  if (r < r1) m0 = __builtin_elementwise_max(m0, *reinterpret_cast<const f32x4*>(src + (size_t)perm[r] * ld + c));
  *reinterpret_cast<f32x4*>(out + (size_t)seg * ld_out + c) = __builtin_elementwise_max(m0, m1);
}

size_t sort_pairs_temp_bytes(int n) {
  size_t b = 0;
  unsigned* k = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, b, k, k, k, k, (size_t)n, 0, 32, (hipStream_t)0);
  return v3d::align_up(b, 256);
}

}  // namespace

// workspace: [sorted keys n*4][iota n*4][rocprim temp]
extern "C" size_t v3d_segment_csr_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return 2 * v3d::align_up((size_t)n * 4, 256) + sort_pairs_temp_bytes(n);
}

extern "C" int v3d_segment_csr(const int32_t* seg_id, int n, int n_seg, int32_t* perm, int32_t* offsets, void* workspace,
                               size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(seg_id && perm && offsets && workspace, V3D_ERR_BAD_ARG, "v3d_segment_csr: null argument");
  V3D_REQUIRE(n > 0 && n_seg > 0, V3D_ERR_BAD_SHAPE, "v3d_segment_csr: n=%d n_seg=%d", n, n_seg);
  V3D_REQUIRE(workspace_bytes >= v3d_segment_csr_workspace_bytes(n), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_segment_csr: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  unsigned* sorted = (unsigned*)base;
  unsigned* iota = (unsigned*)(base + v3d::align_up((size_t)n * 4, 256));
  void* temp = base + 2 * v3d::align_up((size_t)n * 4, 256);
  size_t tb = sort_pairs_temp_bytes(n);
  v3d::TimedScope ts("segment_csr", s);
  iota_kernel<<<(n + 255) / 256, 256, 0, s>>>(iota, n);
  // ids are < n_seg: only the bits that can be set take part in the (stable) sort
  int bits = 1;
  while (bits < 32 && (1ll << bits) < (long long)n_seg) ++bits;
  V3D_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, (const unsigned*)seg_id, sorted, (const unsigned*)iota, (unsigned*)perm,
                                          (size_t)n, 0, bits, s));
  segment_offsets_kernel<<<(n_seg + 1 + 255) / 256, 256, 0, s>>>(sorted, n, n_seg, offsets);
  V3D_CHECK_LAUNCH("segment_offsets_kernel");
  return V3D_OK;
}

extern "C" int v3d_segment_max_f32(const float* src, int ld, const int32_t* perm, const int32_t* offsets, int n_seg, int N,
                                   float* out, int ld_out, void* stream) {
  V3D_REQUIRE(src && perm && offsets && out, V3D_ERR_BAD_ARG, "v3d_segment_max_f32: null argument");
  V3D_REQUIRE(n_seg > 0 && N > 0 && N % 4 == 0 && N <= 256 && ld % 4 == 0 && ld_out % 4 == 0 && ld >= N && ld_out >= N,
              V3D_ERR_BAD_SHAPE, "v3d_segment_max_f32: N=%d ld=%d ld_out=%d (N a multiple of 4, <= 256)", N, ld, ld_out);
  hipStream_t s = (hipStream_t)stream;
  v3d::TimedScope ts("segment_max", s);
  if (N <= 128) segment_max_kernel<32><<<(n_seg + 7) / 8, 256, 0, s>>>(src, ld, perm, offsets, n_seg, N, out, ld_out);
  else segment_max_kernel<64><<<(n_seg + 3) / 4, 256, 0, s>>>(src, ld, perm, offsets, n_seg, N, out, ld_out);
  V3D_CHECK_LAUNCH("segment_max_kernel");
  return V3D_OK;
}
