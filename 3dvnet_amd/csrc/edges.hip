// Edge list -> per-reference CSR on the device (row A7's bookkeeping; reference: mvsnet.py:179 + 214-215,
//   ref_idx, gather_idx = torch.unique(batch.ref_src_edges[0], return_inverse=True);  scatter(x_vox, gather_idx, ...)).
// torch.unique returns a tensor whose length the host has to read back (a device synchronisation per forward, ~0.17 ms
// of idle GPU per 4.5 ms cfg2 step).  The caller of the hot path knows how many reference images a batch holds, so this
// kernel takes that number, builds
//   ref_img  [n_ref]      ascending distinct values of edges[0]                       (= torch.unique's first output)
//   edge_ofs [n_ref + 1]  first edge of every reference in edge_src
//   edge_src [E]          edges[1] grouped by reference, original edge order inside a group (= stable sort by gather_idx)
// without any host round trip, and records in a status word whether the number was right (v3d_edges_csr_status).
// One workgroup: edge lists are a few hundred to a few thousand entries (cfg2: 512, cfg5: 88, a 64-view scene: 512).
#include "v3d_common.h"

namespace {

constexpr int kThreads = 1024, kWaves = kThreads / 64;

struct CsrMeta {
  int magic;
  int error;      // bit 0: number of distinct references != n_ref_expected, bit 1: an image index outside [0, n_img)
  int n_ref;      // distinct references found
  int pad;
};
constexpr int kMagic = 0x43535231;   // "CSR1"
constexpr int kLdsImgs = 4096, kLdsEdges = 8192;

// Inclusive sum over the 1024 threads of the workgroup (wave shuffles + 16 wave totals in LDS: two barriers); `total` = sum of all.
__device__ __forceinline__ int block_inclusive_sum(int v, int* s_wave, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  __syncthreads();                       // s_wave may still be read from the previous call
  if (lane == 63) s_wave[wave] = v;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    const int x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  total = tot;
  return v + base;
}

// workspace: [CsrMeta][rank: n_img ints][count: n_ref_expected ints]
__global__ __launch_bounds__(kThreads) void edges_csr_kernel(const long long* __restrict__ edges, int n_edges, int n_img,
                                                              int n_ref_expected, int* __restrict__ ref_img,
                                                              int* __restrict__ edge_ofs, int* __restrict__ edge_src,
                                                              CsrMeta* __restrict__ meta, int* rank,
                                                              int* __restrict__ count) {
  __shared__ int s_wave[kWaves];
  __shared__ int s_err;
  // small problems (every cost-volume batch) keep the rank table in LDS and cache every edge's reference rank there, so
  // the per-reference passes below never touch global memory; larger ones use the workspace
  __shared__ int s_rank[kLdsImgs];
  __shared__ int s_er[kLdsEdges];
  const bool small = n_img <= kLdsImgs && n_edges <= kLdsEdges;
  if (small) rank = s_rank;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long* const e_ref = edges;
  const long long* const e_src = edges + n_edges;
  if (tid == 0) s_err = 0;
  for (int i = tid; i < n_img; i += kThreads) rank[i] = 0;
  __syncthreads();
  // 1. which images are references
  int bad = 0;
  for (int e = tid; e < n_edges; e += kThreads) {
    const long long r = e_ref[e], s = e_src[e];
    if (r < 0 || r >= n_img || s < 0 || s >= n_img) bad = 1;
    else rank[r] = 1;
  }
  if (bad) s_err = 2;
  __syncthreads();
  // 2. exclusive scan of the flags: rank[i] = index of image i among the references (ascending = torch.unique's order)
  const int per = (n_img + kThreads - 1) / kThreads;
  const int lo = min(tid * per, n_img), hi = min(lo + per, n_img);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += rank[i];
  int n_ref;
  int run = block_inclusive_sum(sum, s_wave, n_ref) - sum;
  for (int i = lo; i < hi; ++i) {
    const int f = rank[i];
    rank[i] = f ? run : -1;
    if (f && run < n_ref_expected) ref_img[run] = i;
    run += f;
  }
  int err = s_err | (n_ref != n_ref_expected ? 1 : 0);
  if (tid == 0) { meta->magic = kMagic; meta->error = err; meta->n_ref = n_ref; }
  __syncthreads();
  if (err) {
    // Nothing downstream may trust the tables.  The consumers read ref_img[r] and that image's camera block for every
    // r < n_ref_expected even when a reference has no edges, so BOTH tables are made safe: no edges anywhere, and every
    // reference slot names image 0 (rows at or above the number of references actually found were never written).
    for (int r = tid; r <= n_ref_expected; r += kThreads) edge_ofs[r] = 0;
    for (int r = tid; r < n_ref_expected; r += kThreads) ref_img[r] = 0;
    return;
  }
  if (small) {
    for (int e = tid; e < n_edges; e += kThreads) s_er[e] = rank[e_ref[e]];
    __syncthreads();
  }
  auto edge_rank = [&](int e) __attribute__((always_inline)) { return small ? s_er[e] : rank[e_ref[e]]; };
  // 3. edges per reference: wave w counts references w, w + 16, ... (ballot + popcount over the edge list)
  for (int r = wave; r < n_ref; r += kWaves) {
    int c = 0;
    for (int e0 = 0; e0 < n_edges; e0 += 64) {
      const int e = e0 + lane;
      const bool hit = e < n_edges && edge_rank(e) == r;
      c += __builtin_popcountll(__ballot(hit));
    }
    if (lane == 0) count[r] = c;
  }
  __syncthreads();
  // 4. offsets
  const int per_r = (n_ref + kThreads - 1) / kThreads;
  const int rlo = min(tid * per_r, n_ref), rhi = min(rlo + per_r, n_ref);
  sum = 0;
  for (int r = rlo; r < rhi; ++r) sum += count[r];
  int n_counted;
  run = block_inclusive_sum(sum, s_wave, n_counted) - sum;
  for (int r = rlo; r < rhi; ++r) { edge_ofs[r] = run; run += count[r]; }
  if (tid == 0) edge_ofs[n_ref] = n_counted;
  __syncthreads();
  // 5. fill, original edge order inside a reference's group (stable)
  for (int r = wave; r < n_ref; r += kWaves) {
    int pos = edge_ofs[r];
    for (int e0 = 0; e0 < n_edges; e0 += 64) {
      const int e = e0 + lane;
      const bool hit = e < n_edges && edge_rank(e) == r;
      const unsigned long long m = __ballot(hit);
      if (hit) edge_src[pos + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)e_src[e];
      pos += __builtin_popcountll(m);
    }
  }
}

}  // namespace

extern "C" size_t v3d_edges_csr_workspace_bytes(int n_img, int n_ref) {
  return sizeof(CsrMeta) + ((size_t)(n_img > 0 ? n_img : 0) + (size_t)(n_ref > 0 ? n_ref : 0)) * sizeof(int);
}

extern "C" int v3d_edges_csr(const int64_t* edges, int n_edges, int n_img, int n_ref, int32_t* ref_img, int32_t* edge_ofs,
                             int32_t* edge_src, void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(edges && ref_img && edge_ofs && edge_src && workspace, V3D_ERR_BAD_ARG, "v3d_edges_csr: null pointer");
  V3D_REQUIRE(n_edges > 0 && n_img > 0 && n_ref > 0 && n_ref <= n_img, V3D_ERR_BAD_SHAPE,
              "v3d_edges_csr: n_edges=%d n_img=%d n_ref=%d", n_edges, n_img, n_ref);
  V3D_REQUIRE(workspace_bytes >= v3d_edges_csr_workspace_bytes(n_img, n_ref), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_edges_csr: workspace %zu < %zu bytes", workspace_bytes, v3d_edges_csr_workspace_bytes(n_img, n_ref));
  hipStream_t s = (hipStream_t)stream;
  CsrMeta* meta = (CsrMeta*)workspace;
  int* rank = (int*)(meta + 1);
  int* count = rank + n_img;
  {
    v3d::TimedScope ts("edges_csr", s);
    edges_csr_kernel<<<1, kThreads, 0, s>>>((const long long*)edges, n_edges, n_img, n_ref, ref_img, edge_ofs, edge_src, meta,
                                            rank, count);
  }
  V3D_CHECK_LAUNCH("edges_csr_kernel");
  return V3D_OK;
}

extern "C" int v3d_edges_csr_status(const void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(workspace && workspace_bytes >= sizeof(CsrMeta), V3D_ERR_BAD_ARG, "v3d_edges_csr_status: not a CSR workspace");
  CsrMeta h;
  V3D_CHECK_HIP(hipMemcpyAsync(&h, workspace, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  V3D_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  V3D_REQUIRE(h.magic == kMagic, V3D_ERR_BAD_ARG, "v3d_edges_csr_status: not a CSR workspace");
  V3D_REQUIRE((h.error & 2) == 0, V3D_ERR_BAD_SHAPE, "v3d_edges_csr: an edge names an image outside [0, n_img)");
  V3D_REQUIRE((h.error & 1) == 0, V3D_ERR_BAD_SHAPE,
              "v3d_edges_csr: the edge list holds %d distinct reference images, not the number the caller passed", h.n_ref);
  return V3D_OK;
}
