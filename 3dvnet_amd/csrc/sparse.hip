// Hash-indexed sparse-tensor structure kernels replacing MinkowskiEngine's coordinate manager on the
// refinement path (semantics restated in SURVEY Appendix A):
//   * open-addressing hash table over packed (batch, x, y, z) coordinates,
//   * 27-offset neighbour ("kernel map") tables for stride-1 / stride-2 convs and stride-2 transposed
//     convs -- consumed as row maps by the gather-GEMM (gemm_gather.hip),
//   * sparse trilinear interpolation at arbitrary query points (MinkowskiInterpolation,
//     mv3d/subnetworks/refinement.py:26,39), written straight into the decoder's wide feature row.
#include "sparse_hash.h"

namespace {

using namespace v3dhash;

__global__ void hash_clear_kernel(HashEntry* entries, unsigned cap, int* status) {
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < cap) entries[i] = HashEntry{kEmpty, -1, 0};
  if (i == 0) *status = 0;
}

__global__ void hash_insert_kernel(HashTable t, const int* __restrict__ coords, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int cb = coords[i * 4], cx = coords[i * 4 + 1], cy = coords[i * 4 + 2], cz = coords[i * 4 + 3];
  if (cb < 0 || cb > 65535 || min(cx, min(cy, cz)) < -kGuard || max(cx, max(cy, cz)) > kCoordMax) {
    atomicOr(t.status, 1);          // would alias another key: reported by v3d_hash_status, row left out
    return;
  }
  const unsigned long long key = pack_key(cb, cx, cy, cz);
  unsigned slot = hash_u64(key) & t.mask;
  for (unsigned probe = 0; probe <= t.mask; ++probe) {
    const unsigned long long prev = atomicCAS(&t.entries[slot].key, kEmpty, key);
    if (prev == kEmpty || prev == key) { t.entries[slot].val = i; return; }   // coordinates are unique rows
    slot = (slot + 1) & t.mask;
  }
}

// nbr[k][p] = row of (out_coords[p] + step * o_k) in the hashed map, o_k in {-1,0,1}^3 with
// k = (ox+1) + 3 (oy+1) + 9 (oz+1);  conv: step = +ts_in, transposed conv: step = -ts_out.
// One thread per (offset, output row): the 27 probes of a row were a loop in one thread -- 27 dependent probe chains in a row,
// 40-50 us per launch whatever the level's size (2.8 k rows took as long as 60 k).  blockIdx.y = offset k, so a wave's stores
// are 64 consecutive ints of column k.
__global__ void neighbors_kernel(HashTable t, const int* __restrict__ out_coords, int n_out, int step,
                                 int* __restrict__ nbr) {
  const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (i >= n_out) return;
  const int4 c = *reinterpret_cast<const int4*>(out_coords + (size_t)i * 4);      // (batch, x, y, z)
  const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
  nbr[(size_t)k * n_out + i] = hash_find(t, pack_key(c.x, c.y + step * ox, c.z + step * oy, c.w + step * oz));
}

// Sparse trilinear interpolation in two kernels:
//  1. corners: one thread per (query, corner) -- the 8 hash probes of a query run in parallel lanes and
//     leave (row, weight) pairs in a small scratch table (row -1 = absent corner, weight unused);
//  2. gather:  C/4 threads per query, float4 channels each; the 8 feature-row loads are independent
//     (no probe in between), so they are all in flight together.
__global__ __launch_bounds__(256) void interp_corners_kernel(HashTable t, int ts, const float* __restrict__ pts,
                                                             const long long* __restrict__ pts_batch, int n_hyp,
                                                             const float* __restrict__ min_pts, float res,
                                                             int n_query, int* __restrict__ crow,
                                                             float* __restrict__ cw) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int q = gid >> 3, corner = gid & 7;
  if (q >= n_query) return;
  const int b = (int)pts_batch[q / n_hyp];
  // query coordinate in base-voxel units: ((p - min) / x.res) * x.stride   (refinement.py:34-35)
  float w = 1.f;
  float cc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float qc = ((pts[(size_t)q * 3 + d] - min_pts[b * 3 + d]) / res) * (float)ts;
    const float c = floorf(qc / (float)ts) * (float)ts + (((corner >> d) & 1) ? ts : 0);
    w *= 1.f - fabsf(qc - c) / ts;
    cc[d] = c;
  }
  int row = -1;
  // coordinates far outside the packed range cannot be present
  if (cc[0] >= -kGuard && cc[1] >= -kGuard && cc[2] >= -kGuard && cc[0] <= 60000.f && cc[1] <= 60000.f && cc[2] <= 60000.f)
    row = hash_find(t, pack_key(b, (int)cc[0], (int)cc[1], (int)cc[2]));
  crow[gid] = row;
  cw[gid] = w;
}

__global__ __launch_bounds__(256) void interp_gather_kernel(const float* __restrict__ feats, int C,
                                                            const int* __restrict__ crow,
                                                            const float* __restrict__ cw, int n_query,
                                                            float* __restrict__ out, int ld_out, int col0) {
  const int tpq = C / 4;                       // threads per query
  const int q = (blockIdx.x * 256 + threadIdx.x) / tpq;
  const int c4 = ((blockIdx.x * 256 + threadIdx.x) % tpq) * 4;
  if (q >= n_query) return;
  int rows[8];
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { rows[k] = crow[q * 8 + k]; w[k] = cw[q * 8 + k]; }
  float4 f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    f[k] = rows[k] >= 0 ? *reinterpret_cast<const float4*>(feats + (size_t)rows[k] * C + c4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {        // corner order x fastest, as in the single-kernel formulation
    if (rows[k] >= 0) { acc.x += w[k] * f[k].x; acc.y += w[k] * f[k].y; acc.z += w[k] * f[k].z; acc.w += w[k] * f[k].w; }
  }
  float* o = out + (size_t)q * ld_out + col0 + c4;
  o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
}

}  // namespace

extern "C" size_t v3d_hash_bytes(int n) { return (size_t)table_capacity(n) * sizeof(HashEntry) + 16; }

extern "C" int v3d_hash_status(const void* table, int n, void* stream) {
  V3D_REQUIRE(table && n > 0, V3D_ERR_BAD_ARG, "v3d_hash_status: bad argument");
  HashTable t = table_view(const_cast<void*>(table), n);
  int st = 0;
  V3D_CHECK_HIP(hipMemcpyAsync(&st, t.status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  V3D_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  V3D_REQUIRE(st == 0, V3D_ERR_BAD_SHAPE,
              "v3d_hash_build: a coordinate is outside the packed-key range (batch 0..65535, x/y/z %d..%d)", -kGuard,
              kCoordMax);
  return V3D_OK;
}

extern "C" int v3d_hash_build(const int32_t* coords, int n, void* table, size_t table_bytes, void* stream) {
  V3D_REQUIRE(coords && table, V3D_ERR_BAD_ARG, "v3d_hash_build: null argument");
  V3D_REQUIRE(n > 0, V3D_ERR_BAD_SHAPE, "v3d_hash_build: empty coordinate map");
  V3D_REQUIRE(table_bytes >= v3d_hash_bytes(n), V3D_ERR_WORKSPACE_TOO_SMALL, "v3d_hash_build: table buffer too small");
  hipStream_t s = (hipStream_t)stream;
  HashTable t = table_view(table, n);
  v3d::TimedScope ts("hash_build", s);
  hash_clear_kernel<<<(t.mask + 256) / 256, 256, 0, s>>>(t.entries, t.mask + 1, t.status);
  hash_insert_kernel<<<(n + 255) / 256, 256, 0, s>>>(t, coords, n);
  V3D_CHECK_LAUNCH("hash_insert_kernel");
  return V3D_OK;
}

extern "C" int v3d_sparse_neighbors(const void* table, int n_in, const int32_t* out_coords, int n_out,
                                    int step, int32_t* nbr, void* stream) {
  V3D_REQUIRE(table && out_coords && nbr, V3D_ERR_BAD_ARG, "v3d_sparse_neighbors: null argument");
  V3D_REQUIRE(n_in > 0 && n_out > 0 && step != 0, V3D_ERR_BAD_SHAPE, "v3d_sparse_neighbors: bad shape");
  V3D_REQUIRE((reinterpret_cast<size_t>(out_coords) & 15) == 0, V3D_ERR_BAD_ARG,
              "v3d_sparse_neighbors: out_coords must be 16-byte aligned (the kernel loads a coordinate row as one int4)");
  hipStream_t s = (hipStream_t)stream;
  HashTable t = table_view(const_cast<void*>(table), n_in);
  v3d::TimedScope ts("sparse_neighbors", s);
  neighbors_kernel<<<dim3((n_out + 255) / 256, 27), 256, 0, s>>>(t, out_coords, n_out, step, nbr);
  V3D_CHECK_LAUNCH("neighbors_kernel");
  return V3D_OK;
}

extern "C" size_t v3d_sparse_interp_workspace_bytes(int n_pts, int n_hyp) {
  return v3d::align_up((size_t)n_pts * n_hyp * 8 * 4, 256) * 2;
}

extern "C" int v3d_sparse_interp_f32(const void* table, int n_in, const float* feats, int C,
                                     int tensor_stride, const float* pts, const int64_t* pts_batch,
                                     int n_pts, int n_hyp, const float* min_pts, float res, float* out,
                                     int ld_out, int col0, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  V3D_REQUIRE(table && feats && pts && pts_batch && min_pts && out, V3D_ERR_BAD_ARG,
              "v3d_sparse_interp_f32: null argument");
  V3D_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024 && 256 % (C / 4) == 0, V3D_ERR_UNSUPPORTED,
              "v3d_sparse_interp_f32: C=%d unsupported", C);
  V3D_REQUIRE(n_in > 0 && n_pts >= 0 && n_hyp > 0 && tensor_stride > 0 && res > 0.f, V3D_ERR_BAD_SHAPE,
              "v3d_sparse_interp_f32: bad shape");
  const long long nq = (long long)n_pts * n_hyp;
  if (nq == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  HashTable t = table_view(const_cast<void*>(table), n_in);
  V3D_REQUIRE(workspace && workspace_bytes >= v3d_sparse_interp_workspace_bytes(n_pts, n_hyp),
              V3D_ERR_WORKSPACE_TOO_SMALL, "v3d_sparse_interp_f32: workspace too small");
  V3D_REQUIRE(nq * 8 < (1ll << 31), V3D_ERR_BAD_SHAPE, "v3d_sparse_interp_f32: too many queries");
  int* crow = (int*)workspace;
  float* cw = (float*)((char*)workspace + v3d::align_up((size_t)nq * 8 * 4, 256));
  const long long threads = nq * (C / 4);
  v3d::TimedScope ts("sparse_interp", s);
  interp_corners_kernel<<<(unsigned)((nq * 8 + 255) / 256), 256, 0, s>>>(
      t, tensor_stride, pts, (const long long*)pts_batch, n_hyp, min_pts, res, (int)nq, crow, cw);
  interp_gather_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(feats, C, crow, cw, (int)nq, out, ld_out, col0);
  V3D_CHECK_LAUNCH("interp_gather_kernel");
  return V3D_OK;
}
