// Row B3 of SURVEY.md §8a and the coordinate-map part of row B6: voxelisation of the scene point cloud
// (mv3d/utils.py:38-64 + torch_geometric voxel_grid / torch_cluster grid, restated) and the output
// coordinate map of a stride-2 sparse convolution (unique(floor(c / 2ts) * 2ts), lexicographic order).
// Sorting / de-duplication of the 64-bit keys uses rocPRIM's device radix sort and unique; everything
// else (bounding box, voxel ids, inverse map, decode, per-batch shift) is a small kernel here.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "v3d_common.h"

namespace {

struct VoxMeta {          // device-resident, produced by bbox_finish_kernel
  float bmin[3], bmax[3];
  long long grid[3];      // ceil((max - min) / edge)            (utils.py:41)  -- used to DECODE
  long long num[4];       // trunc((end - start) / size) + 1     (torch_cluster) -- used to ENCODE
  long long cum[4];
  long long max_batch;
  int error;              // != 0: the scene does not fit the fixed tables (v3d_voxelize_status)
};

constexpr int kMaxBatch = 1024;        // rows of the per-batch min-index table in the workspace
constexpr long long kMaxGrid = 65000;  // voxel indices must fit the 16-bit fields of the sparse-tensor keys (sparse.hip)

constexpr int kRedBlocks = 256;

__global__ __launch_bounds__(256) void bbox_partial_kernel(const float* __restrict__ pts,
                                                           const long long* __restrict__ batch, int n,
                                                           float* __restrict__ part) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float mb = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = pts[(size_t)i * 3 + d];
      lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v);
      if (!(fabsf(v) <= 3e38f)) mb = INFINITY;        // NaN / inf coordinate: poisons the batch maximum -> error word
    }
    const long long bi = batch[i];
    mb = fmaxf(mb, bi < 0 ? INFINITY : (float)bi);
  }
  __shared__ float s[7][256];
#pragma unroll
  for (int d = 0; d < 3; ++d) { s[d][threadIdx.x] = lo[d]; s[3 + d][threadIdx.x] = hi[d]; }
  s[6][threadIdx.x] = mb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        s[d][threadIdx.x] = fminf(s[d][threadIdx.x], s[d][threadIdx.x + o]);
        s[3 + d][threadIdx.x] = fmaxf(s[3 + d][threadIdx.x], s[3 + d][threadIdx.x + o]);
      }
      s[6][threadIdx.x] = fmaxf(s[6][threadIdx.x], s[6][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 7) part[blockIdx.x * 7 + threadIdx.x] = s[threadIdx.x][0];
}

// one wave: the partials are spread over the 64 lanes and reduced with shuffles (one thread walking 256 partials was 256
// dependent round trips: 50 us for a 7-word result); minima / maxima do not depend on the order
__global__ void bbox_finish_kernel(const float* __restrict__ part, int nblocks, float edge, VoxMeta* meta) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, mb = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], part[b * 7 + d]); hi[d] = fmaxf(hi[d], part[b * 7 + 3 + d]); }
    mb = fmaxf(mb, part[b * 7 + 6]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o)); }
    mb = fmaxf(mb, __shfl_xor(mb, o));
  }
  if (threadIdx.x != 0) return;
  long long cum = 1;
  for (int d = 0; d < 3; ++d) {
    meta->bmin[d] = lo[d]; meta->bmax[d] = hi[d];
    meta->grid[d] = (long long)ceilf((hi[d] - lo[d]) / edge);
    meta->num[d] = (long long)((hi[d] - lo[d]) / edge) + 1;        // trunc toward zero
    meta->cum[d] = cum; cum *= meta->num[d];
  }
  int err = 0;
  if (!(mb >= 0.f) || !(mb < (float)kMaxBatch)) { err |= 1; mb = mb < 3e18f ? mb : -1.f; }
  meta->num[3] = (long long)mb + 1;                                 // batch dimension: size 1, start 0
  meta->cum[3] = cum;
  meta->max_batch = (long long)mb;
  for (int d = 0; d < 3; ++d)
    if (!(hi[d] >= lo[d]) || meta->grid[d] > kMaxGrid || meta->num[d] > kMaxGrid) err |= 2;
  meta->error = err;
}

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float* __restrict__ pts,
                                                         const long long* __restrict__ batch, int n, float edge,
                                                         const VoxMeta* __restrict__ meta,
                                                         unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long id = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) id += (long long)((pts[(size_t)i * 3 + d] - meta->bmin[d]) / edge) * meta->cum[d];
  id += (long long)((float)batch[i]) * meta->cum[3];                // (batch - 0) / 1 in float, truncated
  keys[i] = (unsigned long long)id;
}

__global__ __launch_bounds__(256) void lower_bound_kernel(const unsigned long long* __restrict__ sorted_unique,
                                                          int n_u, const unsigned long long* __restrict__ q,
                                                          int n, long long* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = q[i];
  int lo = 0, hi = n_u;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_unique[mid] < key) lo = mid + 1; else hi = mid; }
  out[i] = lo;
}

__global__ __launch_bounds__(256) void voxel_decode_kernel(const unsigned long long* __restrict__ uniq, int n_u,
                                                           float edge, float half_edge,
                                                           const VoxMeta* __restrict__ meta,
                                                           float* __restrict__ anchor_pts, int* __restrict__ idx3d,
                                                           long long* __restrict__ anchor_batch,
                                                           int* __restrict__ min_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n_u;
  const long long id = live ? (long long)uniq[i] : 0;
  if (meta->error) return;                                    // reported by v3d_voxelize_status; nothing is written
  const long long b = id / meta->cum[3];                      // == scatter-min of pts_batch over the voxel (:50)
  const long long gxy = meta->grid[0] * meta->grid[1];
  const long long rem = id - b * (gxy * meta->grid[2]);       // anchor_idx -= anchor_batch * max_grid_idx (:53)
  const int z = (int)(rem / gxy);                             // (:55-57), int32 like the reference tensor
  const int y = (int)((rem - (long long)z * gxy) / meta->grid[0]);
  const int x = (int)((rem - (long long)z * gxy) % meta->grid[0]);
  const int c[3] = {x, y, z};
  // scatter-min per batch (:61): keys are sorted, so a wave almost always holds one batch -> reduce in the
  // wave and issue one atomic per wave and coordinate instead of one per voxel
  const int b0 = __shfl((int)b, 0);
  const bool uniform = __all(!live || (int)b == b0);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (live) {
      idx3d[i * 3 + d] = c[d];
      // idx * edge_len + bbox_min + edge_len / 2 with one rounding per operation, as the tensor ops (:58)
      anchor_pts[i * 3 + d] = v3d::add_rn(v3d::add_rn(v3d::mul_rn((float)c[d], edge), meta->bmin[d]), half_edge);
    }
    if (uniform) {
      int m = live ? c[d] : 0x7fffffff;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
      if ((threadIdx.x & 63) == 0 && m != 0x7fffffff) atomicMin(&min_idx[b0 * 3 + d], m);
    } else if (live) {
      atomicMin(&min_idx[b * 3 + d], c[d]);
    }
  }
  if (live) anchor_batch[i] = b;
}

__global__ __launch_bounds__(256) void voxel_shift_kernel(int* __restrict__ idx3d, const long long* __restrict__ ab,
                                                          const int* __restrict__ min_idx, int n_u,
                                                          const VoxMeta* __restrict__ meta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_u * 3 || meta->error) return;
  idx3d[i] -= min_idx[ab[i / 3] * 3 + i % 3];                                   // (:62)
}

__global__ void fill_int_kernel(int* p, int n, int v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

constexpr int kGuard = 8;
__global__ __launch_bounds__(256) void strided_keys_kernel(const int* __restrict__ coords, int n, int ts2,
                                                           unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long k = (unsigned long long)(unsigned)(coords[i * 4] & 0xffff) << 48;
#pragma unroll
  for (int d = 1; d < 4; ++d) {
    const int c = coords[i * 4 + d];
    const int f = (c >= 0 ? c / ts2 : -((-c + ts2 - 1) / ts2)) * ts2;           // floor(c / ts2) * ts2
    k |= (unsigned long long)(unsigned)((f + kGuard) & 0xffff) << (16 * (3 - d));
  }
  keys[i] = k;
}

__global__ __launch_bounds__(256) void unpack_keys_kernel(const unsigned long long* __restrict__ keys, int n,
                                                          int* __restrict__ coords) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  coords[i * 4] = (int)(k >> 48);
#pragma unroll
  for (int d = 1; d < 4; ++d) coords[i * 4 + d] = (int)((k >> (16 * (3 - d))) & 0xffff) - kGuard;
}

size_t sort_unique_temp_bytes(int n) {
  size_t b1 = 0, b2 = 0;
  unsigned long long* p = nullptr;
  unsigned int* c = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, b1, p, p, (size_t)n, 0, 64, (hipStream_t)0);
  (void)rocprim::unique(nullptr, b2, p, p, c, (size_t)n, rocprim::equal_to<unsigned long long>(), (hipStream_t)0);
  return v3d::align_up(b1 > b2 ? b1 : b2, 256);
}

}  // namespace

// workspace layout of v3d_sort_unique_u64: [sorted keys n*8][count 256][rocprim temp]
extern "C" size_t v3d_sort_unique_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return v3d::align_up((size_t)n * 8, 256) + 256 + sort_unique_temp_bytes(n);
}

extern "C" int v3d_sort_unique_u64(const uint64_t* keys_in, int n, uint64_t* keys_out, int* n_unique_host,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(keys_in && keys_out && n_unique_host && workspace, V3D_ERR_BAD_ARG, "v3d_sort_unique_u64: null argument");
  V3D_REQUIRE(n > 0, V3D_ERR_BAD_SHAPE, "v3d_sort_unique_u64: n must be positive");
  V3D_REQUIRE(workspace_bytes >= v3d_sort_unique_workspace_bytes(n), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_sort_unique_u64: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  unsigned long long* sorted = (unsigned long long*)base;
  unsigned int* count = (unsigned int*)(base + v3d::align_up((size_t)n * 8, 256));
  void* temp = base + v3d::align_up((size_t)n * 8, 256) + 256;
  size_t tb = sort_unique_temp_bytes(n);
  {
    v3d::TimedScope ts("sort_unique", s);
    V3D_CHECK_HIP(rocprim::radix_sort_keys(temp, tb, (const unsigned long long*)keys_in, sorted, (size_t)n, 0, 64, s));
    V3D_CHECK_HIP(rocprim::unique(temp, tb, sorted, (unsigned long long*)keys_out, count, (size_t)n,
                                  rocprim::equal_to<unsigned long long>(), s));
  }
  unsigned int host_count = 0;
  // the output size is data dependent: this call returns it and therefore synchronises the stream
  V3D_CHECK_HIP(hipMemcpyAsync(&host_count, count, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  V3D_CHECK_HIP(hipStreamSynchronize(s));
  *n_unique_host = (int)host_count;
  return V3D_OK;
}

extern "C" int v3d_strided_keys(const int32_t* coords, int n, int tensor_stride, uint64_t* keys_out, void* stream) {
  V3D_REQUIRE(coords && keys_out, V3D_ERR_BAD_ARG, "v3d_strided_keys: null argument");
  V3D_REQUIRE(n > 0 && tensor_stride > 0, V3D_ERR_BAD_SHAPE, "v3d_strided_keys: bad shape");
  strided_keys_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(coords, n, 2 * tensor_stride,
                                                                       (unsigned long long*)keys_out);
  V3D_CHECK_LAUNCH("strided_keys_kernel");
  return V3D_OK;
}

extern "C" int v3d_unpack_coords(const uint64_t* keys, int n, int32_t* coords_out, void* stream) {
  V3D_REQUIRE(keys && coords_out, V3D_ERR_BAD_ARG, "v3d_unpack_coords: null argument");
  V3D_REQUIRE(n > 0, V3D_ERR_BAD_SHAPE, "v3d_unpack_coords: bad shape");
  unpack_keys_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((const unsigned long long*)keys, n, coords_out);
  V3D_CHECK_LAUNCH("unpack_keys_kernel");
  return V3D_OK;
}

// workspace layout of the voxelize calls: [VoxMeta 256][partials kRedBlocks*7 floats][min_idx 1024*3 ints]
extern "C" size_t v3d_voxelize_workspace_bytes(void) {
  return 256 + v3d::align_up(kRedBlocks * 7 * sizeof(float), 256) + kMaxBatch * 3 * sizeof(int);
}

// Host-visible result of the range checks made by v3d_voxel_keys (synchronises the stream; the Python caller runs it
// right after v3d_sort_unique_u64, which has synchronised already).
extern "C" int v3d_voxelize_status(const void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(workspace && workspace_bytes >= v3d_voxelize_workspace_bytes(), V3D_ERR_BAD_ARG,
              "v3d_voxelize_status: not a voxelize workspace");
  VoxMeta m;
  V3D_CHECK_HIP(hipMemcpyAsync(&m, workspace, sizeof(VoxMeta), hipMemcpyDeviceToHost, (hipStream_t)stream));
  V3D_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  V3D_REQUIRE((m.error & 1) == 0, V3D_ERR_BAD_SHAPE, "voxelize: batch id %lld outside [0, %d) (-1: negative id or non-finite point)",
              m.max_batch, kMaxBatch);
  V3D_REQUIRE((m.error & 2) == 0, V3D_ERR_BAD_SHAPE,
              "voxelize: grid %lld x %lld x %lld exceeds %lld cells per axis (or the point cloud holds NaN)", m.grid[0],
              m.grid[1], m.grid[2], kMaxGrid);
  return V3D_OK;
}

extern "C" int v3d_voxel_keys(const float* pts, const int64_t* pts_batch, int n, float edge_len,
                              uint64_t* keys_out, void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(pts && pts_batch && keys_out && workspace, V3D_ERR_BAD_ARG, "v3d_voxel_keys: null argument");
  V3D_REQUIRE(n > 0 && edge_len > 0.f, V3D_ERR_BAD_SHAPE, "v3d_voxel_keys: bad shape");
  V3D_REQUIRE(workspace_bytes >= v3d_voxelize_workspace_bytes(), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_voxel_keys: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  VoxMeta* meta = (VoxMeta*)workspace;
  float* part = (float*)((char*)workspace + 256);
  const int nb = min(kRedBlocks, (n + 255) / 256);
  v3d::TimedScope ts("voxel_keys", s);
  bbox_partial_kernel<<<nb, 256, 0, s>>>(pts, (const long long*)pts_batch, n, part);
  bbox_finish_kernel<<<1, 64, 0, s>>>(part, nb, edge_len, meta);
  voxel_keys_kernel<<<(n + 255) / 256, 256, 0, s>>>(pts, (const long long*)pts_batch, n, edge_len, meta,
                                                    (unsigned long long*)keys_out);
  V3D_CHECK_LAUNCH("voxel_keys_kernel");
  return V3D_OK;
}

extern "C" int v3d_lower_bound_u64(const uint64_t* sorted_unique, int n_unique, const uint64_t* queries, int n,
                                   int64_t* index_out, void* stream) {
  V3D_REQUIRE(sorted_unique && queries && index_out, V3D_ERR_BAD_ARG, "v3d_lower_bound_u64: null argument");
  V3D_REQUIRE(n_unique > 0 && n > 0, V3D_ERR_BAD_SHAPE, "v3d_lower_bound_u64: bad shape");
  lower_bound_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
      (const unsigned long long*)sorted_unique, n_unique, (const unsigned long long*)queries, n, (long long*)index_out);
  V3D_CHECK_LAUNCH("lower_bound_kernel");
  return V3D_OK;
}

extern "C" int v3d_voxel_decode(const uint64_t* unique_keys, int n_unique, float edge_len, float half_edge,
                                float* anchor_pts, int32_t* anchor_idx3d, int64_t* anchor_batch,
                                void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(unique_keys && anchor_pts && anchor_idx3d && anchor_batch && workspace, V3D_ERR_BAD_ARG,
              "v3d_voxel_decode: null argument");
  V3D_REQUIRE(n_unique > 0, V3D_ERR_BAD_SHAPE, "v3d_voxel_decode: bad shape");
  V3D_REQUIRE(workspace_bytes >= v3d_voxelize_workspace_bytes(), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_voxel_decode: workspace too small (must be the buffer v3d_voxel_keys filled)");
  hipStream_t s = (hipStream_t)stream;
  const VoxMeta* meta = (const VoxMeta*)workspace;
  int* min_idx = (int*)((char*)workspace + 256 + v3d::align_up(kRedBlocks * 7 * sizeof(float), 256));
  v3d::TimedScope ts("voxel_decode", s);
  fill_int_kernel<<<(kMaxBatch * 3 + 255) / 256, 256, 0, s>>>(min_idx, kMaxBatch * 3, 0x7fffffff);
  voxel_decode_kernel<<<(n_unique + 255) / 256, 256, 0, s>>>((const unsigned long long*)unique_keys, n_unique,
                                                             edge_len, half_edge, meta, anchor_pts, anchor_idx3d,
                                                             (long long*)anchor_batch, min_idx);
  voxel_shift_kernel<<<(n_unique * 3 + 255) / 256, 256, 0, s>>>(anchor_idx3d, (const long long*)anchor_batch,
                                                               min_idx, n_unique, meta);
  V3D_CHECK_LAUNCH("voxel_shift_kernel");
  return V3D_OK;
}
