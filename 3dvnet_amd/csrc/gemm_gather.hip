// Gather-GEMM on the exact-fp32 matrix cores: the dense-arithmetic workhorse of rows B4 (PointNet),
// B6 (sparse 3D U-Net) and C2b (hypothesis decoder conv1d) of SURVEY.md §8a.
//
//   Y[m, :] = epilogue( sum_{s < n_seg}  act(X_s[row_s(m), 0:K]) @ W_s  + bias )
//
// A "segment" is (source matrix, row map, weight slab):
//   * PointNet fcK(relu(cat(x, pool[idx])))  -> 2 segments: (x, identity), (pool, idx)
//                                                (scenemodeling.py:129-141)
//   * sparse conv, 27 kernel offsets           -> 27 segments over one source, row map = column k of
//                                                the neighbour table, -1 = absent voxel = zero row
//                                                (MinkowskiConvolution, scenemodeling.py:36-38,160,181)
//   * conv1d(k3, pad 1) along the 7 hypotheses -> 3 segments, row map m + (s - 1) inside each group
//                                                of `group_len` rows (refinement.py:8-25)
// Tile: 128 rows x N (<= 128) outputs per 256-thread workgroup.  D[co, row] orientation: the A operand
// is the weight fragment (pre-packed, staged per 32-wide K chunk into LDS), the B operand the
// gathered activations (LDS, row stride 34 floats = conflict-free for the 4 k-lanes x 16 row-lanes).
// Each wave owns all 8 row blocks x N/64 channel blocks, so the per-row GroupNorm (16-channel groups
// == one MFMA channel block, scenemodeling.py:98-104) needs only two cross-lane shuffles.
// Segments whose row map is empty for the whole tile are skipped (structured sparsity).
// Epilogue (all optional): + bias, GroupNorm(rows), + residual, ReLU, scatter-max into pool[idx[m]]
// (order-independent => deterministic), row-major store.
#include <cstdlib>
#include <vector>

#include "v3d_common.h"
#include "gemm_weights.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));


// hi = RNE_bf16(x), lo = RNE_bf16(x - hi) of four floats on packed pairs: v_cvt_pk_bf16_f32 rounds like bf16_rne for finite values
// (conv0z.hip uses the same pair) at 12 instead of ~50 vector instructions per four values -- the split was the largest item of the
// staging threads' time in the sparse convolutions
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
__device__ __forceinline__ u32x2 split_hi(f32x4 v) { return (u32x2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}; }
__device__ __forceinline__ u32x2 split_lo(f32x4 v, u32x2 h) {
  return (u32x2){pack_bf16x2(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u)),
                 pack_bf16x2(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u))};
}

constexpr int kMaxSeg = 27;
constexpr int kKC = 32;      // K chunk
constexpr int kXS = kKC + 2; // LDS row stride of the activation tile

struct Seg {
  const float* src;
  const int* idx;   // row map or null (identity)
  int ld;           // row stride of src in floats
};

struct GemmParams {
  Seg seg[kMaxSeg];
  int n_seg, M, N, K, KP;        // K real columns per segment, KP = K rounded up to kKC
  int group_len;                 // > 0: conv1d row map (segment s reads row m + s - n_seg/2 of its group)
  int relu_in;
  const float* wp;               // packed [seg][KP/32][8][MB][64]
  const float* bias;             // [N] or null
  const float* gn_w; const float* gn_b; float gn_eps;   // row GroupNorm over 16-channel groups (null = off)
  int gn8;                                              // ... over 8-channel groups instead (use_gn == 8)
  const float* residual; int ld_res;
  int relu_out;
  float* pool; const int* pool_idx; int ld_pool;        // scatter-max target (null = off)
  float* out; int ld_out;                               // null = do not store
};

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  // order-independent max for mixed-sign floats; target initialised to -inf
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// MBW: channel blocks (16 outputs) per wave (N <= 64 * MBW); NB: 16-row blocks per workgroup tile
// (tile = 16 * NB rows: 128 for large problems, 32 when M is small so that more workgroups than CUs
// exist and gather latency is hidden by occupancy).
// BF16: operands are split x = hi + lo into two bf16 values (16 mantissa bits together) and a product block is
// evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- 3 bf16 MFMAs
// (K = 32) instead of 8 fp32 MFMAs (K = 4), i.e. 5.3x the matrix rate at ~7e-6 relative layer error (the
// depth gate is 1e-4).  The activation tile is then [row][32 k] bf16 (64 B rows, hi and lo
// arrays) with the 16-B k-group slot XOR-swizzled by the row so that every ds_read_b128 lane group hits 16
// distinct slots; weight fragments are split and packed on the host.
// ---- epilogue shared by the GEMM kernels: bias -> GroupNorm(16) -> residual -> ReLU -> scatter-max -> store -----------
template <int MBW, int NB>
__device__ __forceinline__ void gemm_epilogue_generic(const GemmParams& p, f32x4 (&acc)[NB][MBW], int m0, int wave, int kq, int jn) {
  // ---- epilogue: lane (kq, jn) holds channels co0 .. co0+3 of row m0 + nb*16 + jn -------------------
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int m = m0 + nb * 16 + jn;
#pragma unroll
    for (int mw = 0; mw < MBW; ++mw) {
      const int co0 = (wave * MBW + mw) * 16 + kq * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[nb][mw][r] + ((p.bias && co0 + r < p.N) ? p.bias[co0 + r] : 0.f);
      if (p.gn_w) {
        // GroupNorm over the 16 channels of this block for row m: 4 registers x 4 lane quarters (8-channel groups --
        // SparseUNet(dims[0] = 32, 4 groups), the reference's feat_dim = 16 -- are the lane quarter pairs (0, 1) and (2, 3))
        const float ginv = p.gn8 ? 1.f / 8.f : 1.f / 16.f;
        float sum = v[0] + v[1] + v[2] + v[3];
        sum += __shfl_xor(sum, 16);
        if (!p.gn8) sum += __shfl_xor(sum, 32);
        const float mean = sum * ginv;
        float sq = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sq += (v[r] - mean) * (v[r] - mean);
        sq += __shfl_xor(sq, 16);
        if (!p.gn8) sq += __shfl_xor(sq, 32);
        const float rstd = 1.f / sqrtf(sq * ginv + p.gn_eps);      // biased variance (torch GN)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co0 + r < p.N) v[r] = (v[r] - mean) * rstd * p.gn_w[co0 + r] + p.gn_b[co0 + r];
      }
      if (m < p.M && co0 < p.N) {
        if (p.residual) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co0 + r < p.N) v[r] += p.residual[(size_t)m * p.ld_res + co0 + r];
        }
        if (p.relu_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.pool) {
          float* pr = p.pool + (size_t)p.pool_idx[m] * p.ld_pool + co0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co0 + r < p.N) atomic_max_float(pr + r, v[r]);
        }
        if (p.out) {
          float* o = p.out + (size_t)m * p.ld_out + co0;
          if (co0 + 3 < p.N && (p.ld_out % 4 == 0)) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co0 + r < p.N) o[r] = v[r];
          }
        }
      }
    }
  }
}

// The same epilogue for the shapes every hot caller has (N a multiple of 16, 16-byte aligned rows of `out` / `residual`, no
// scatter-max): the generic code above loads bias / GroupNorm affine / residual one float at a time inside the (row block,
// channel block) loops and waits for each -- the compiler cannot move a load across the stores of the previous block -- which made
// the epilogue 28 k of the 46 k cycles of a PointNet layer's workgroup (scripts/micro/phase_dense_gemm.py) and ~10 % of a sparse
// convolution.  Here the per-channel constants are read once as float4, the residual rows of the whole tile are requested before
// the first is used, and nothing but the stores is predicated.  Same arithmetic, same order: bit-identical.
template <int MBW, int NB>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[NB][MBW], int m0, int wave, int kq, int jn) {
  const bool fast = p.N % 16 == 0 && !p.pool && (!p.out || (p.ld_out % 4 == 0 && (reinterpret_cast<size_t>(p.out) & 15) == 0)) &&
                    (!p.residual || (p.ld_res % 4 == 0 && (reinterpret_cast<size_t>(p.residual) & 15) == 0)) &&
                    (!p.bias || (reinterpret_cast<size_t>(p.bias) & 15) == 0) &&
                    (!p.gn_w || ((reinterpret_cast<size_t>(p.gn_w) & 15) == 0 && (reinterpret_cast<size_t>(p.gn_b) & 15) == 0));
  if (!fast) {
    gemm_epilogue_generic<MBW, NB>(p, acc, m0, wave, kq, jn);
    return;
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 b4[MBW], gw4[MBW], gb4[MBW];
  int coc[MBW];
  bool cok[MBW];
#pragma unroll
  for (int mw = 0; mw < MBW; ++mw) {
    const int co0 = (wave * MBW + mw) * 16 + kq * 4;
    cok[mw] = co0 < p.N;
    coc[mw] = cok[mw] ? co0 : 0;                    // a channel block behind N reads block 0 and stores nothing
    b4[mw] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + coc[mw]) : zero4;
    if (!cok[mw]) b4[mw] = zero4;                   // (the generic code adds no bias there; the GroupNorm of such a block is never stored)
    gw4[mw] = p.gn_w ? *reinterpret_cast<const f32x4*>(p.gn_w + coc[mw]) : zero4;
    gb4[mw] = p.gn_w ? *reinterpret_cast<const f32x4*>(p.gn_b + coc[mw]) : zero4;
  }
  // the residual rows of four row blocks at a time are requested before the first is used (all of a 64-row tile)
  constexpr int G = NB < 4 ? NB : 4;
#pragma unroll
  for (int g0 = 0; g0 < NB; g0 += G) {
    f32x4 res[G][MBW];
    if (p.residual) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int m = m0 + (g0 + g) * 16 + jn, mc = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int mw = 0; mw < MBW; ++mw)
          res[g][mw] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mc * p.ld_res + coc[mw]);
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int nb = g0 + g, m = m0 + nb * 16 + jn;
#pragma unroll
      for (int mw = 0; mw < MBW; ++mw) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nb][mw][r] + b4[mw][r];
        if (p.gn_w) {
          const float ginv = p.gn8 ? 1.f / 8.f : 1.f / 16.f;
          float sum = v[0] + v[1] + v[2] + v[3];
          sum += __shfl_xor(sum, 16);
          if (!p.gn8) sum += __shfl_xor(sum, 32);
          const float mean = sum * ginv;
          float sq = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) sq += (v[r] - mean) * (v[r] - mean);
          sq += __shfl_xor(sq, 16);
          if (!p.gn8) sq += __shfl_xor(sq, 32);
          const float rstd = 1.f / sqrtf(sq * ginv + p.gn_eps);      // biased variance (torch GN)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (v[r] - mean) * rstd * gw4[mw][r] + gb4[mw][r];
        }
        if (p.residual) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += res[g][mw][r];
        }
        if (p.relu_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.out && m < p.M && cok[mw])
          *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.ld_out + coc[mw]) = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
  }
}

#ifdef V3D_PHASE_TIMING
// developer build only: per-phase cycle counters (wave 0 of every workgroup), see costreg.hip
constexpr int kPhaseSlots = 1 << 16;
__device__ unsigned long long g_gg_phase[8 * kPhaseSlots];
#define PHASE_DECL                                  \
  long long ph_t = __builtin_readcyclecounter();    \
  long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_MARK(i)                                   \
  do {                                                  \
    long long t_ = __builtin_readcyclecounter();        \
    ph_acc[i] += t_ - ph_t;                             \
    ph_t = t_;                                          \
  } while (0)
#define PHASE_FLUSH                                                                                      \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < kPhaseSlots)                                                    \
      for (int i_ = 0; i_ < 8; ++i_) g_gg_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_];   \
  } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_FLUSH
#endif

template <int MBW, int NB, bool BF16>
__global__ __launch_bounds__(256) void gemm_gather_kernel(GemmParams p) {
  constexpr int kTM = 16 * NB;
  constexpr int NPASS = (kTM + 31) / 32;      // staging passes of 32 rows
  __shared__ __attribute__((aligned(16))) float xs[kTM * kXS];
  __shared__ __attribute__((aligned(16))) float ws[4 * MBW * 16 * kKC];
  __shared__ int s_act[kMaxSeg], s_list[kMaxSeg], s_nact;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  const int m0 = blockIdx.x * kTM;
  const int MB = 4 * MBW;

  f32x4 acc[NB][MBW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int m = 0; m < MBW; ++m) acc[nb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging role: 8 lanes x float4 cover the 32 columns of one row; 32 rows per pass
  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  const int nkc = p.KP / kKC;
  constexpr int kWslab = 4 * MBW * 16 * kKC;       // packed floats per (segment, K chunk)
  constexpr int WPT = kWslab / 4 / 256;            // float4 of the weight slab per thread
  static_assert(kWslab % 1024 == 0, "weight slab must split evenly over the workgroup");

  // row source of this thread's staging rows for segment s (-1 = zero row)
  auto rows_of = [&](int s, int (&r)[NPASS]) __attribute__((always_inline)) {
    const int* idx = p.seg[s].idx;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int m = m0 + srow + 32 * i;
      int v = -1;
      if (m < p.M && srow + 32 * i < kTM) {
        if (p.group_len > 0) {
          const int h = m % p.group_len + s - p.n_seg / 2;
          v = (h >= 0 && h < p.group_len) ? m + s - p.n_seg / 2 : -1;
        } else {
          v = idx ? idx[m] : m;
        }
      }
      r[i] = v;
    }
  };

  // ---- which segments does this tile need at all? (structured sparsity of the neighbour tables) ---------
  if (tid < kMaxSeg) s_act[tid] = 0;
  __syncthreads();
  for (int s = 0; s < p.n_seg; ++s) {
    bool any = p.seg[s].idx == nullptr;            // identity / conv1d maps are always needed
    if (!any) {
      int r[NPASS];
      rows_of(s, r);
#pragma unroll
      for (int i = 0; i < NPASS; ++i) any |= r[i] >= 0;
    }
    if (any) s_act[s] = 1;                         // benign race: everybody writes the same value
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int s = 0; s < p.n_seg; ++s) if (s_act[s]) s_list[n++] = s;
    s_nact = n;
  }
  __syncthreads();
  const int nact = s_nact;

  // ---- software pipeline over the flat (active segment, K chunk) sequence: the next chunk's gathered rows
  // and weight slab are loaded into registers right after the barrier and land during the MFMA loop -------
  f32x4 xr[NPASS], wr[WPT];     // native vector type: keeps the staging registers out of scratch
  auto issue = [&](int s, int kc, const int (&r)[NPASS]) __attribute__((always_inline)) {
    const Seg sg = p.seg[s];
    const bool vec_ok = (sg.ld % 4 == 0) && (p.K % 4 == 0) && ((reinterpret_cast<size_t>(sg.src) & 15) == 0);
    const int col = kc * kKC + sc4;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (r[i] >= 0 && col < p.K) {
        const float* rowp = sg.src + (size_t)r[i] * sg.ld + col;
        if (vec_ok) {
          v = *reinterpret_cast<const f32x4*>(rowp);
        } else {
          v.x = rowp[0];
          if (col + 1 < p.K) v.y = rowp[1];
          if (col + 2 < p.K) v.z = rowp[2];
          if (col + 3 < p.K) v.w = rowp[3];
        }
      }
      xr[i] = v;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wp + (size_t)(s * nkc + kc) * kWslab) + tid;
#pragma unroll
    for (int i = 0; i < WPT; ++i) wr[i] = wsrc[i * 256];
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      if (srow + 32 * i < kTM) {
        f32x4 v = xr[i];
        if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const int row = srow + 32 * i;
        if constexpr (BF16) {
          // this thread holds k = sc4 .. sc4+3 of the row: half of the 8-wide k group kg = sc4 / 8
          const int kg = sc4 >> 3, half = (sc4 >> 2) & 1;
          const int slot = kg ^ ((((row & 15) >> 3) & 1) * 3);
          const u32x2 hp = split_hi(v), lp = split_lo(v, hp);
          u32x2* xh2 = reinterpret_cast<u32x2*>(xs);
          xh2[(row * 4 + slot) * 2 + half] = hp;
          xh2[((kTM + row) * 4 + slot) * 2 + half] = lp;
        } else {
          float* d = xs + row * kXS + sc4;
          *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
          *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) reinterpret_cast<f32x4*>(ws)[tid + i * 256] = wr[i];
  };

  PHASE_DECL;
  int rcur[NPASS], rnxt[NPASS];
  if (nact > 0) {
    rows_of(s_list[0], rcur);
    issue(s_list[0], 0, rcur);
    if (nact > 1) rows_of(s_list[1], rnxt);
  }
  PHASE_MARK(0);
  for (int a = 0; a < nact; ++a) {
    for (int kc = 0; kc < nkc; ++kc) {
      __syncthreads();
      PHASE_MARK(1);
      commit();
      PHASE_MARK(2);
      __syncthreads();
      PHASE_MARK(3);
      if (kc + 1 < nkc) {
        issue(s_list[a], kc + 1, rcur);
      } else if (a + 1 < nact) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) rcur[i] = rnxt[i];
        issue(s_list[a + 1], 0, rcur);
        if (a + 2 < nact) rows_of(s_list[a + 2], rnxt);       // one whole segment ahead of its first use
      }
      PHASE_MARK(4);
      // ---- MFMA ------------------------------------------------------------------------------------------
      if constexpr (BF16) {
        const u32x4* wq = reinterpret_cast<const u32x4*>(ws);
        const u32x4* xq = reinterpret_cast<const u32x4*>(xs);
        const int bslot = kq ^ (((jn >> 3) & 1) * 3);
        bf16x8 a_hi[MBW], a_lo[MBW];
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
          a_hi[m] = __builtin_bit_cast(bf16x8, wq[(wave * MBW + m) * 64 + lane]);
          a_lo[m] = __builtin_bit_cast(bf16x8, wq[(MB + wave * MBW + m) * 64 + lane]);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xq[(nb * 16 + jn) * 4 + bslot]);
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xq[(kTM + nb * 16 + jn) * 4 + bslot]);
#pragma unroll
          for (int m = 0; m < MBW; ++m) {
            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[m], b_hi, acc[nb][m], 0, 0, 0);
            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[m], b_lo, acc[nb][m], 0, 0, 0);
            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[m], b_hi, acc[nb][m], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int k4 = 0; k4 < kKC / 4; ++k4) {
          float a_frag[MBW];
#pragma unroll
          for (int m = 0; m < MBW; ++m) a_frag[m] = ws[((k4 * MB + wave * MBW + m) * 64) + lane];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float bv = xs[(nb * 16 + jn) * kXS + k4 * 4 + kq];
#pragma unroll
            for (int m = 0; m < MBW; ++m)
              acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_frag[m], bv, acc[nb][m], 0, 0, 0);
          }
        }
      }
    }
  }

  PHASE_MARK(5);
  gemm_epilogue<MBW, NB>(p, acc, m0, wave, kq, jn);
  PHASE_MARK(6);
  PHASE_FLUSH;
}

// ---- the same GEMM in rounds of S (segment, K chunk) steps: the small-M kernel of the sparse convolutions ------------------
// gemm_gather_kernel moves ONE step per pair of barriers: a thread has one 16-byte gather (and its share of the weight slab)
// in flight, so a tile's time is (number of steps) x (a memory round trip): 27 offsets x K / 32 chunks -- 108 steps on the
// 128-channel level of the sparse U-Net, whose 3.5 k rows are 110 workgroups on 256 CUs (nothing else hides the latency).
// Here a round stages S steps' activation tiles at once (S gathers per thread in flight, one pair of barriers per round) and the
// weight fragments go from L2 straight into the registers of the wave that owns the channel block (they were never shared
// between waves; the LDS round trip of the slab is gone).  The neighbour-table column of every segment is read once into an
// LDS row table (the old kernel read it twice: activity scan + staging).  Same step order and MFMA order per accumulator:
// bit-identical to gemm_gather_kernel<MBW, NB, true>.
template <int MBW, int NB, int S>
__global__ __launch_bounds__(256) void gemm_gather_rounds_kernel(GemmParams p) {
  constexpr int kTM = 16 * NB, NPASS = (kTM + 31) / 32, MB = 4 * MBW;
  constexpr int kWslab = 4 * MBW * 16 * kKC;       // packed floats per (segment, K chunk)
  constexpr int kTile = 2 * kTM * 4;                // 16-byte slots of one step's activation tile: [hi, lo][row][4 k groups]
  __shared__ __attribute__((aligned(16))) u32x4 xq[S * kTile];
  __shared__ int rowtab[kMaxSeg * kTM];
  __shared__ int s_act[kMaxSeg], s_list[kMaxSeg], s_nact;
  __shared__ int steptab[kMaxSeg * 8 + 2 * S];     // (segment << 8 | K chunk) of every step of the flat sequence, -1 behind its end

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  // rows are sorted by voxel key: a contiguous run of tiles per XCD keeps the neighbour rows a tile gathers in that XCD's L2
  const int m0 = v3d::xcd_contiguous_block() * kTM;
  const int nkc = p.KP / kKC;

  f32x4 acc[NB][MBW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int m = 0; m < MBW; ++m) acc[nb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- row table: source row of (segment, tile row), -1 = zero row; which segments does the tile need at all? ----------------
  if (tid < kMaxSeg) s_act[tid] = 0;
  __syncthreads();
  for (int e = tid; e < p.n_seg * kTM; e += 256) {
    const int sg = e / kTM, row = e % kTM, m = m0 + row;
    int v = -1;
    if (m < p.M) {
      if (p.group_len > 0) {
        const int h = m % p.group_len + sg - p.n_seg / 2;
        v = (h >= 0 && h < p.group_len) ? m + sg - p.n_seg / 2 : -1;
      } else {
        v = p.seg[sg].idx ? p.seg[sg].idx[m] : m;
      }
    }
    rowtab[e] = v;
    if (v >= 0 || p.seg[sg].idx == nullptr) s_act[sg] = 1;       // benign race: everybody writes the same value
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int sg = 0; sg < p.n_seg; ++sg) if (s_act[sg]) s_list[n++] = sg;
    s_nact = n;
  }
  __syncthreads();
  const int n_steps = s_nact * nkc;                 // flat (active segment, K chunk) sequence
  for (int f = tid; f < n_steps + 2 * S; f += 256) steptab[f] = f < n_steps ? (s_list[f / nkc] << 8) | (f % nkc) : -1;
  __syncthreads();
  // One source matrix for every segment (a sparse convolution: only the row maps differ): its pointer and row stride are read
  // once.  Otherwise they are looked up per step -- a scalar load from the kernel arguments behind an LDS read, ~500 cycles in
  // front of every gather (measured: 1.9 k cycles per round of four steps just to issue the gathers).
  bool one_src = true;
  for (int sg = 1; sg < p.n_seg; ++sg) one_src = one_src && p.seg[sg].src == p.seg[0].src && p.seg[sg].ld == p.seg[0].ld;
  const float* const src0 = p.seg[0].src;
  const int ld0 = p.seg[0].ld;
  PHASE_DECL;

  // staging role: 8 lanes x float4 cover the 32 columns of one row; 32 rows per pass
  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  f32x4 xr[S][NPASS];
  u32x4 afr[S][2 * MBW];
  int st[S];                                        // the steps of the round being issued (wave-uniform)
  auto load_steps = [&](int f0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < S; ++j) st[j] = __builtin_amdgcn_readfirstlane(steptab[f0 + j]);
  };
  auto issue_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (st[j] >= 0) {
        const int sg = st[j] >> 8, col = (st[j] & 255) * kKC + sc4;
        const float* const src = one_src ? src0 : p.seg[sg].src;
        const int ld = one_src ? ld0 : p.seg[sg].ld;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
          const int row = srow + 32 * i;
          const int r = row < kTM ? rowtab[sg * kTM + row] : -1;
          xr[j][i] = (r >= 0 && col < p.K) ? *reinterpret_cast<const f32x4*>(src + (size_t)r * ld + col)
                                           : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto issue_a = [&](int stj, u32x4 (&a)[2 * MBW]) __attribute__((always_inline)) {
    if (stj >= 0) {
      const int sg = stj >> 8, kc = stj & 255;
      const u32x4* w = reinterpret_cast<const u32x4*>(p.wp + (size_t)(sg * nkc + kc) * kWslab) + lane;
#pragma unroll
      for (int m = 0; m < MBW; ++m) {
        a[m] = w[(wave * MBW + m) * 64];
        a[MBW + m] = w[(MB + wave * MBW + m) * 64];
      }
    }
  };
  auto commit = [&](int f0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (f0 + j < n_steps) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
          const int row = srow + 32 * i;
          if (row < kTM) {
            f32x4 v = xr[j][i];
            if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            // this thread holds k = sc4 .. sc4+3 of the row: half of the 8-wide k group kg = sc4 / 8
            const int kg = sc4 >> 3, half = (sc4 >> 2) & 1;
            const int slot = kg ^ ((((row & 15) >> 3) & 1) * 3);
            const u32x2 hp = split_hi(v), lp = split_lo(v, hp);
            u32x2* xh2 = reinterpret_cast<u32x2*>(xq + j * kTile);
            xh2[(row * 4 + slot) * 2 + half] = hp;
            xh2[((kTM + row) * 4 + slot) * 2 + half] = lp;
          }
        }
      }
    }
  };

  load_steps(0);
  issue_x();
#pragma unroll
  for (int j = 0; j < S; ++j) issue_a(st[j], afr[j]);
  const int bslot = kq ^ (((jn >> 3) & 1) * 3);
  PHASE_MARK(0);
#pragma unroll 1
  for (int f0 = 0; f0 < n_steps; f0 += S) {
    __syncthreads();                 // the previous round's MFMAs are done with the activation tiles
    PHASE_MARK(1);
    commit(f0);
    PHASE_MARK(2);
    __syncthreads();
    PHASE_MARK(3);
    load_steps(f0 + S);
    issue_x();                       // the next round's gathers fly during this round's MFMAs
    PHASE_MARK(4);
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (f0 + j < n_steps) {
        const u32x4* xs = xq + j * kTile;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xs[(nb * 16 + jn) * 4 + bslot]);
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xs[(kTM + nb * 16 + jn) * 4 + bslot]);
#pragma unroll
          for (int m = 0; m < MBW; ++m) {
            const bf16x8 a_hi = __builtin_bit_cast(bf16x8, afr[j][m]), a_lo = __builtin_bit_cast(bf16x8, afr[j][MBW + m]);
            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc[nb][m], 0, 0, 0);
            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc[nb][m], 0, 0, 0);
            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc[nb][m], 0, 0, 0);
          }
        }
      }
      issue_a(st[j], afr[j]);        // this step's fragment registers are free: the next round's step j takes them
    }
    PHASE_MARK(5);
  }
  gemm_epilogue<MBW, NB>(p, acc, m0, wave, kq, jn);
  PHASE_MARK(6);
  PHASE_FLUSH;
}

// ---- the sparse convolution as a loader / matrix pipeline (round 5) ----------------------------------------------------------
// Phase counters of the rounds kernel on the cfg3 U-Net (13 434 rows x 128 channels, 32-row tiles): of 154 k cycles per
// workgroup 38 k wait for the gathers, 49 k ISSUE them (the CU's vector-memory path is busy with the 1.77 MB of weight
// fragments every tile streams: 64 KB per round) and 49 k are the matrix instructions + the wait for those fragments --
// nothing overlaps, because one wave does all three in turn and its loads retire in order.  Here the roles are separate waves
// with separate load queues: the loader waves (4 or 8 of them behind the four matrix waves) gather the neighbour rows U rounds
// ahead (registers), split them and fill a 2-slot LDS ring; waves 0..3 keep the weight fragments of the next AD - 1 rounds in
// flight and run the matrix instructions; one barrier per round of S steps.  64-row tiles halve the weight stream per row (one workgroup per CU at 13 k rows).  Every load is
// unconditional (absent rows read row 0 and are masked in registers; rounds behind the end repeat step 0 and are not
// multiplied), so the compiler's s_waitcnt counting is exact and the prefetch distance survives.  Same step order and MFMA
// order per accumulator as gemm_gather_kernel<MBW, NB, true>: bit-identical.
// Step sequence of a tile: the active segments in ascending order x the K chunks, kept as a bit mask + a chunk counter in
// scalar registers (an LDS step table costs an LDS round trip + v_readfirstlane in front of every load address).
struct StepIter {
  unsigned mask;   // active segments not yet finished (wave-uniform)
  int kc;
  __device__ __forceinline__ bool next(int nkc, int& sg, int& k) {   // false behind the end (sg = k = 0 then)
    const bool ok = mask != 0;
    sg = ok ? __builtin_ctz(mask) : 0;
    k = ok ? kc : 0;
    const bool wrap = kc + 1 == nkc;
    mask = (ok && wrap) ? (mask & (mask - 1)) : mask;
    kc = ok ? (wrap ? 0 : kc + 1) : kc;
    return ok;
  }
};

#ifndef V3D_PIPE_ABLATE
#define V3D_PIPE_ABLATE 0     // developer ablations (scripts/micro/sparse_ablate.sh): 1 no MFMAs, 2 no fragment loads, 3 no gathers, 4 no split / LDS writes
#endif
// NL: loader waves (4 or 8); AD: depth of the weight-fragment ring of the matrix waves (fragments of AD - 1 rounds in flight)
template <int MBW, int NB, int NL, int AD, int MINB>
__global__ __launch_bounds__(256 + 64 * NL, MINB) void gemm_gather_pipe_kernel(GemmParams p) {
  constexpr int kTM = 16 * NB, MB = 4 * MBW, S = 2, U = 4, NT = 256 + 64 * NL;
  constexpr int RPP = NL * 8;                       // rows per loader pass: 8 threads x 16 B = one 128-byte row slice
  constexpr int NPASS = (kTM + RPP - 1) / RPP;
  constexpr int kWslab = 4 * MBW * 16 * kKC;        // packed floats per (segment, K chunk)
  constexpr int kTile = 2 * kTM * 4;                // 16-byte slots of one step's activation tile: [hi, lo][row][4 k groups]
  static_assert(U % AD == 0 && AD >= 2, "the fragment ring must divide the unroll");
  __shared__ __attribute__((aligned(16))) u32x4 xq[2 * S * kTile];
  __shared__ int rowtab[kMaxSeg * kTM];
  __shared__ unsigned s_mask;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = v3d::xcd_contiguous_block() * kTM;
  const int nkc = p.KP / kKC;

  if (tid == 0) s_mask = 0;
  __syncthreads();
  {
    unsigned mine = 0;
    for (int e = tid; e < p.n_seg * kTM; e += NT) {
      const int sg = e / kTM, row = e % kTM, m = m0 + row;
      const int v = m < p.M ? p.seg[sg].idx[m] : -1;
      rowtab[e] = v;
      mine |= (v >= 0 ? 1u : 0u) << sg;
    }
    if (mine) atomicOr(&s_mask, mine);
  }
  __syncthreads();
  const unsigned act = (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask);
  const int n_steps = __builtin_popcount(act) * nkc;
  const int n_rounds = (n_steps + S - 1) / S, n_iter = (n_rounds + U - 1) / U * U;
  const float* const src = p.seg[0].src;
  const int ld = p.seg[0].ld;

  if (wave >= 4) {
    // ---- loader waves ---------------------------------------------------------------------------------------------------------
    const int ltid = tid - 256, srow = ltid >> 3, sc4 = (ltid & 7) * 4;
    f32x4 xr[U][S][NPASS];
    unsigned vm[U];
    StepIter it = {act, 0};
    auto issue = [&](f32x4 (&x)[S][NPASS], unsigned& valid) __attribute__((always_inline)) {
      valid = 0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        int sg, kc;
        const bool live = it.next(nkc, sg, kc);
        const int col = kc * kKC + sc4;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
          const int row = srow + RPP * i;
          const int r = row < kTM ? rowtab[sg * kTM + row] : -1;
          const bool ok = live && r >= 0 && col < p.K;
          valid |= (ok ? 1u : 0u) << (j * NPASS + i);
          const unsigned ofs = (unsigned)(r < 0 ? 0 : r) * (unsigned)ld + (unsigned)(col < p.K ? col : 0);
#if V3D_PIPE_ABLATE == 3
          x[j][i] = (f32x4){(float)ofs, 0.f, 0.f, 0.f};
#else
          x[j][i] = *reinterpret_cast<const f32x4*>(src + ofs);
#endif
        }
      }
    };
    auto commit = [&](int slot, const f32x4 (&x)[S][NPASS], unsigned valid) __attribute__((always_inline)) {
#if V3D_PIPE_ABLATE == 4
#pragma unroll
      for (int j = 0; j < S; ++j)
#pragma unroll
        for (int i = 0; i < NPASS; ++i) asm volatile("" ::"v"(x[j][i]));
      return;
#endif
#pragma unroll
      for (int j = 0; j < S; ++j) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
          const int row = srow + RPP * i;
          if (kTM % RPP == 0 || row < kTM) {
            f32x4 v = x[j][i];
            const unsigned keep = 0u - ((valid >> (j * NPASS + i)) & 1u);      // zero row: absent neighbour / behind the end
            v.x = __uint_as_float(__float_as_uint(v.x) & keep); v.y = __uint_as_float(__float_as_uint(v.y) & keep);
            v.z = __uint_as_float(__float_as_uint(v.z) & keep); v.w = __uint_as_float(__float_as_uint(v.w) & keep);
            if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const int kg = sc4 >> 3, half = (sc4 >> 2) & 1;
            const int sl = kg ^ ((((row & 15) >> 3) & 1) * 3);
            const u32x2 hp = split_hi(v), lp = split_lo(v, hp);
            u32x2* xh2 = reinterpret_cast<u32x2*>(xq + (slot * S + j) * kTile);
            xh2[(row * 4 + sl) * 2 + half] = hp;
            xh2[((kTM + row) * 4 + sl) * 2 + half] = lp;
          }
        }
      }
    };
    PHASE_DECL;
#pragma unroll
    for (int u = 0; u < U; ++u) issue(xr[u], vm[u]);
    commit(0, xr[0], vm[0]);
    issue(xr[0], vm[0]);
    __syncthreads();
    PHASE_MARK(7);
#pragma unroll 1
    for (int r0 = 0; r0 < n_iter; r0 += U) {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        commit((i + 1) & 1, xr[(i + 1) % U], vm[(i + 1) % U]);       // round r + 1 -> the slot the matrix waves left a round ago
        PHASE_MARK(4);
        issue(xr[(i + 1) % U], vm[(i + 1) % U]);                     // round r + 1 + U
        PHASE_MARK(5);
        __syncthreads();
        PHASE_MARK(6);
      }
    }
#ifdef V3D_PHASE_TIMING
    if (threadIdx.x == 256 && blockIdx.x < kPhaseSlots)
      for (int i_ = 4; i_ < 8; ++i_) g_gg_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_];
#endif
    return;
  }

  // ---- matrix waves ---------------------------------------------------------------------------------------------------------------
  const int kq = lane >> 4, jn = lane & 15;
  f32x4 acc[NB][MBW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int m = 0; m < MBW; ++m) acc[nb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 afr[AD][S][2 * MBW];
  StepIter ita = {act, 0};
  const u32x4* const wbase = reinterpret_cast<const u32x4*>(p.wp) + wave * MBW * 64 + lane;
  auto issue_a = [&](u32x4 (&a)[S][2 * MBW]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      int sg, kc;
      ita.next(nkc, sg, kc);
      const u32x4* w = wbase + (size_t)(sg * nkc + kc) * (kWslab / 4);
#pragma unroll
      for (int m = 0; m < MBW; ++m) {
#if V3D_PIPE_ABLATE == 2
        a[j][m] = (u32x4){(unsigned)(size_t)w, 1u, 2u, 3u};
        a[j][MBW + m] = (u32x4){(unsigned)(size_t)w, 1u, 2u, 3u};
#else
        a[j][m] = w[m * 64];
        a[j][MBW + m] = w[(MB + m) * 64];
#endif
      }
    }
  };
  PHASE_DECL;
#pragma unroll
  for (int u = 0; u < AD - 1; ++u) issue_a(afr[u]);
  const int bslot = kq ^ (((jn >> 3) & 1) * 3);
  __syncthreads();
  PHASE_MARK(0);
#pragma unroll 1
  for (int r0 = 0; r0 < n_iter; r0 += U) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int r = r0 + i;
      issue_a(afr[(i + AD - 1) % AD]);                               // round r + AD - 1
      PHASE_MARK(1);
#pragma unroll
      for (int j = 0; j < S; ++j) {
        if (r * S + j < n_steps) {
          const u32x4* xs = xq + ((i & 1) * S + j) * kTile;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xs[(nb * 16 + jn) * 4 + bslot]);
            const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xs[(kTM + nb * 16 + jn) * 4 + bslot]);
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
              const bf16x8 a_hi = __builtin_bit_cast(bf16x8, afr[i % AD][j][m]), a_lo = __builtin_bit_cast(bf16x8, afr[i % AD][j][MBW + m]);
#if V3D_PIPE_ABLATE == 1
              asm volatile("" ::"v"(a_hi), "v"(a_lo), "v"(b_hi), "v"(b_lo));
#else
              acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc[nb][m], 0, 0, 0);
              acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc[nb][m], 0, 0, 0);
              acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc[nb][m], 0, 0, 0);
#endif
            }
          }
        }
      }
      PHASE_MARK(2);
      __syncthreads();
      PHASE_MARK(3);
    }
  }
  gemm_epilogue<MBW, NB>(p, acc, m0, wave, kq, jn);
#ifdef V3D_PHASE_TIMING
  if (threadIdx.x == 0 && blockIdx.x < kPhaseSlots)
    for (int i_ = 0; i_ < 4; ++i_) g_gg_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_];
#endif
}

// ---- Conv1d(k = 3, pad 1) over groups of `group_len` consecutive rows (the hypothesis decoder, refinement.py:29-30) ----
// The three taps read the same activation rows shifted by -1 / 0 / +1, so the tile (+ one halo row on either side) is
// gathered, split and committed to LDS once per K chunk and used by all three taps; only the 16 KB weight slab changes
// between taps.  A tap that would cross a group boundary reads a zero row instead (the conv's padding).  The generic
// kernel stages every (tap, chunk) separately: three times the gather + split work, which is the larger half of its time.
template <int MBW, int NB>
__global__ __launch_bounds__(256) void conv1d_gemm_kernel(GemmParams p) {
  constexpr int kTM = 16 * NB, kZRow = kTM + 2, kRT = kTM + 3;      // rows in LDS: halo + tile + halo + zero row
  constexpr int NPASS = (kTM + 2 + 31) / 32;
  constexpr int kWslab = 4 * MBW * 16 * kKC, MB = 4 * MBW;
  __shared__ __attribute__((aligned(16))) u32x4 xq[2 * kRT * 4];     // [hi, lo][row][4 k groups of 8 bf16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  const int m0 = blockIdx.x * kTM;
  const int nkc = p.KP / kKC;
  const float* const src = p.seg[0].src;
  const int ld = p.seg[0].ld;

  f32x4 acc[NB][MBW];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int m = 0; m < MBW; ++m) acc[nb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // per column block: does tap 0 / tap 2 of this lane's output row stay inside its group?  (bit 2 nb / 2 nb + 1; a tap
  // that leaves the group reads the zero row = the conv's padding)
  unsigned tapmask = 0;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int hh = (m0 + nb * 16 + jn) % p.group_len;
    tapmask |= (hh >= 1 ? 1u : 0u) << (2 * nb);
    tapmask |= (hh + 1 < p.group_len ? 1u : 0u) << (2 * nb + 1);
  }
  if (tid < 8) xq[(tid >> 2) * kRT * 4 + kZRow * 4 + (tid & 3)] = (u32x4){0u, 0u, 0u, 0u};

  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  f32x4 xr[NPASS];
  auto issue_x = [&](int kc) __attribute__((always_inline)) {
    const int col = kc * kKC + sc4;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int lr = srow + 32 * i, m = m0 - 1 + lr;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (lr < kTM + 2 && m >= 0 && m < p.M && col < p.K) v = *reinterpret_cast<const f32x4*>(src + (size_t)m * ld + col);
      xr[i] = v;
    }
  };
  auto commit_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = srow + 32 * i;
      if (row < kTM + 2) {
        f32x4 v = xr[i];
        if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const int kg = sc4 >> 3, half = (sc4 >> 2) & 1;
        const int slot = kg ^ (((row >> 3) & 1) * 3);
        const u32x2 hp = split_hi(v), lp = split_lo(v, hp);
        u32x2* xh2 = reinterpret_cast<u32x2*>(xq);
        xh2[(row * 4 + slot) * 2 + half] = hp;
        xh2[((kRT + row) * 4 + slot) * 2 + half] = lp;
      }
    }
  };
  // A fragments: every wave owns its MBW channel blocks, nothing is shared between waves, so the fragments go from the
  // packed image (L2-resident) straight into registers, one tap ahead -- no LDS copy, no barrier between taps
  u32x4 a_cur[2 * MBW], a_nxt[2 * MBW];
  auto load_a = [&](u32x4 (&a)[2 * MBW], int t, int kc) __attribute__((always_inline)) {
    const u32x4* w = reinterpret_cast<const u32x4*>(p.wp + (size_t)(t * nkc + kc) * kWslab) + lane;
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
      a[m] = w[(wave * MBW + m) * 64];
      a[MBW + m] = w[(MB + wave * MBW + m) * 64];
    }
  };

  issue_x(0);
  load_a(a_cur, 0, 0);
#pragma unroll 1
  for (int kc = 0; kc < nkc; ++kc) {
    __syncthreads();                 // the previous chunk's MFMAs are done with the tile
    commit_x();
    __syncthreads();
    if (kc + 1 < nkc) issue_x(kc + 1);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < 2) load_a(a_nxt, t + 1, kc);
      else if (kc + 1 < nkc) load_a(a_nxt, 0, kc + 1);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bool inside = t == 1 || ((tapmask >> (2 * nb + (t >> 1))) & 1u);
        const int R = inside ? nb * 16 + jn + t : kZRow;
        const int slot = R * 4 + (kq ^ (((R >> 3) & 1) * 3));
        const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xq[slot]);
        const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xq[kRT * 4 + slot]);
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
          const bf16x8 a_hi = __builtin_bit_cast(bf16x8, a_cur[m]), a_lo = __builtin_bit_cast(bf16x8, a_cur[MBW + m]);
          acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc[nb][m], 0, 0, 0);
          acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc[nb][m], 0, 0, 0);
          acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc[nb][m], 0, 0, 0);
        }
      }
#pragma unroll
      for (int m = 0; m < 2 * MBW; ++m) a_cur[m] = a_nxt[m];
    }
  }
  gemm_epilogue<MBW, NB>(p, acc, m0, wave, kq, jn);
}


__global__ void fill_kernel(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

// ---- packed weights -------------------------------------------------------------------------------

extern "C" int v3d_gemm_pack(const float* w_host, long long stride_seg, long long stride_co,
                             long long stride_k, int n_seg, int N, int K, const float* scale_host,
                             const float* bias_host, const float* gn_w_host, const float* gn_b_host,
                             v3d_gemm_weights** out_handle) {
  V3D_REQUIRE(w_host && out_handle, V3D_ERR_BAD_ARG, "v3d_gemm_pack: null argument");
  V3D_REQUIRE(n_seg >= 1 && n_seg <= kMaxSeg && N >= 1 && N <= 128 && K >= 1, V3D_ERR_BAD_SHAPE,
              "v3d_gemm_pack: unsupported shape (n_seg=%d N=%d K=%d)", n_seg, N, K);
  v3d_gemm_weights* h = new v3d_gemm_weights();
  h->N = N; h->K = K; h->n_seg = n_seg;
  h->KP = (K + kKC - 1) / kKC * kKC;
  h->MBW = N > 64 ? 2 : 1;
  const int MB = 4 * h->MBW, nkc = h->KP / kKC;
  const size_t wslab = (size_t)MB * 16 * kKC;
  std::vector<float> host((size_t)n_seg * nkc * wslab + 3 * 128, 0.f);   // grown below for the bf16 image
  for (int s = 0; s < n_seg; ++s)
    for (int kc = 0; kc < nkc; ++kc)
      for (int k4 = 0; k4 < kKC / 4; ++k4)
        for (int mb = 0; mb < MB; ++mb)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = mb * 16 + (lane & 15), k = kc * kKC + k4 * 4 + (lane >> 4);
            float v = 0.f;
            if (co < N && k < K) {
              v = w_host[s * stride_seg + co * stride_co + k * stride_k];
              if (scale_host) v *= scale_host[co];
            }
            host[((size_t)(s * nkc + kc) * wslab) + ((size_t)(k4 * MB + mb) * 64) + lane] = v;
          }
  h->bias_ofs = (size_t)n_seg * nkc * wslab;
  h->gnw_ofs = h->bias_ofs + 128;
  h->gnb_ofs = h->gnw_ofs + 128;
  // split-bf16 image: per (segment, chunk): [hi, lo][MB][64 lanes][4 words]; lane l holds output co = mb*16 + (l & 15),
  // k = chunk*32 + 8*(l >> 4) + e (e = 0..7, two bf16 per word)
  h->bf_ofs = h->gnb_ofs + 128;
  host.resize(h->bf_ofs + (size_t)n_seg * nkc * wslab, 0.f);
  {
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->bf_ofs);
    auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int s = 0; s < n_seg; ++s)
      for (int kc = 0; kc < nkc; ++kc)
        for (int mb = 0; mb < MB; ++mb)
          for (int lane = 0; lane < 64; ++lane) {
            unsigned hi[8], lo[8];
            for (int e = 0; e < 8; ++e) {
              const int co = mb * 16 + (lane & 15), k = kc * kKC + 8 * (lane >> 4) + e;
              float v = 0.f;
              if (co < N && k < K) {
                v = w_host[s * stride_seg + co * stride_co + k * stride_k];
                if (scale_host) v *= scale_host[co];
              }
              hi[e] = rne(v);
              lo[e] = rne(v - up(hi[e]));
            }
            for (int part = 0; part < 2; ++part) {
              const unsigned* src = part ? lo : hi;
              unsigned* dst = wb + (size_t)(s * nkc + kc) * wslab + ((size_t)(part * MB + mb) * 64 + lane) * 4;
              for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
            }
          }
  }
  h->dec_ofs = 0;
  if (n_seg == 3 && N == 128 && K % 16 == 0) {
    const int nst = K / 16;
    h->dec_ofs = host.size();
    host.resize(h->dec_ofs + (size_t)nst * 24 * 256, 0.f);
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->dec_ofs);
    auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int st = 0; st < nst; ++st)
      for (int t = 0; t < 3; ++t)
        for (int mb = 0; mb < 4; ++mb)
          for (int lane = 0; lane < 64; ++lane) {
            unsigned hi[8], lo[8];
            const int co = mb * 32 + (lane & 31), g = lane >> 5;
            for (int e = 0; e < 8; ++e) {
              const int k = 16 * st + 8 * (e >> 2) + 4 * g + (e & 3);
              float v = w_host[t * stride_seg + co * stride_co + k * stride_k];
              if (scale_host) v *= scale_host[co];
              hi[e] = rne(v);
              lo[e] = rne(v - up(hi[e]));
            }
            for (int part = 0; part < 2; ++part) {
              const unsigned* src = part ? lo : hi;
              unsigned* dst = wb + ((((size_t)st * 3 + t) * 2 + part) * 4 + mb) * 256 + (size_t)lane * 4;
              for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
            }
          }
  }
  h->has_bias = bias_host != nullptr;
  h->has_gn = gn_w_host != nullptr && gn_b_host != nullptr;
  for (int i = 0; i < N; ++i) {
    if (bias_host) host[h->bias_ofs + i] = bias_host[i];
    if (h->has_gn) { host[h->gnw_ofs + i] = gn_w_host[i]; host[h->gnb_ofs + i] = gn_b_host[i]; }
  }
  hipError_t e = hipMalloc((void**)&h->dev, host.size() * sizeof(float));
  if (e != hipSuccess) { delete h; return v3d::fail(V3D_ERR_HIP, "hipMalloc(gemm weights): %s", hipGetErrorString(e)); }
  e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return v3d::fail(V3D_ERR_HIP, "hipMemcpy(gemm weights): %s", hipGetErrorString(e)); }
  *out_handle = h;
  return V3D_OK;
}

extern "C" void v3d_gemm_free(v3d_gemm_weights* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  delete h;
}

extern "C" int v3d_gemm_gather_f32(const v3d_gemm_weights* h, int M, const float* const* seg_src_host,
                                   const int32_t* const* seg_idx_host, const int* seg_ld_host,
                                   int group_len, int relu_in, int use_gn, float gn_eps,
                                   const float* residual, int ld_res, int relu_out, float* pool,
                                   const int32_t* pool_idx, int ld_pool, float* out, int ld_out,
                                   int precision, void* stream) {
  V3D_REQUIRE(h && seg_src_host && seg_ld_host, V3D_ERR_BAD_ARG, "v3d_gemm_gather_f32: null argument");
  V3D_REQUIRE(precision == V3D_PRECISION_SPLIT_BF16 || precision == V3D_PRECISION_FP32, V3D_ERR_BAD_ARG,
              "v3d_gemm_gather_f32: unknown precision %d", precision);
  V3D_REQUIRE(M >= 0, V3D_ERR_BAD_SHAPE, "v3d_gemm_gather_f32: M < 0");
  V3D_REQUIRE(use_gn == 0 || use_gn == 1 || use_gn == 8 || use_gn == 16, V3D_ERR_BAD_ARG,
              "v3d_gemm_gather_f32: use_gn = %d (0 off, 1 or 16: 16-channel groups, 8: 8-channel groups)", use_gn);
  V3D_REQUIRE(!use_gn || (h->has_gn && h->N % 16 == 0), V3D_ERR_BAD_ARG,
              "v3d_gemm_gather_f32: GroupNorm requested but weights carry no affine / N %% 16 != 0");
  V3D_REQUIRE(!pool || pool_idx, V3D_ERR_BAD_ARG, "v3d_gemm_gather_f32: pool without pool_idx");
  if (M == 0) return V3D_OK;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  for (int s = 0; s < h->n_seg; ++s) {
    V3D_REQUIRE(seg_src_host[s], V3D_ERR_BAD_ARG, "v3d_gemm_gather_f32: segment %d has no source", s);
    p.seg[s].src = seg_src_host[s];
    p.seg[s].idx = seg_idx_host ? seg_idx_host[s] : nullptr;
    p.seg[s].ld = seg_ld_host[s];
  }
  p.n_seg = h->n_seg; p.M = M; p.N = h->N; p.K = h->K; p.KP = h->KP;
  p.group_len = group_len; p.relu_in = relu_in;
  p.wp = h->dev; p.bias = h->has_bias ? h->dev + h->bias_ofs : nullptr;
  p.gn_w = use_gn ? h->dev + h->gnw_ofs : nullptr; p.gn_b = use_gn ? h->dev + h->gnb_ofs : nullptr;
  p.gn_eps = gn_eps; p.gn8 = use_gn == 8;
  p.residual = residual; p.ld_res = ld_res; p.relu_out = relu_out;
  p.pool = pool; p.pool_idx = pool_idx; p.ld_pool = ld_pool;
  p.out = out; p.ld_out = ld_out;
  hipStream_t s = (hipStream_t)stream;
  const bool small = M < 128 * 1024;           // fewer than ~4 tiles of 128 rows per CU: use 32-row tiles
  const int tm = small ? 32 : 128;
  const unsigned blocks = (unsigned)((M + tm - 1) / tm);
  const bool fp32_path = precision == V3D_PRECISION_FP32;    // exact-fp32 MFMA instead of split bf16
  if (!fp32_path) p.wp = h->dev + h->bf_ofs;
  // conv1d over row groups (3 taps of the same source): the tile is staged once per K chunk for all three taps
  bool conv1d = !fp32_path && group_len > 0 && h->n_seg == 3 && h->K % 4 == 0;
  for (int t = 0; conv1d && t < 3; ++t)
    conv1d = p.seg[t].src == p.seg[0].src && p.seg[t].ld == p.seg[0].ld && !p.seg[t].idx && p.seg[t].ld % 4 == 0 &&
             (reinterpret_cast<size_t>(p.seg[t].src) & 15) == 0;
  if (conv1d) {
    v3d::TimedScope ts("conv1d_gemm", s);
    // one kernel family for every M (32-row tiles when M is small): chunked and unchunked calls of the decoder then
    // accumulate in the same order and agree bit for bit
    // 64-row tiles for large M: 32 accumulator registers instead of 64 -> 4 waves per SIMD (14.3 ms against 16.0 ms with
    // 128-row tiles and 17.9 ms with 32-row tiles on the cfg3 decoder)
    const unsigned blocks64 = (unsigned)((M + 63) / 64);
    if (h->MBW == 2) { if (small) conv1d_gemm_kernel<2, 2><<<blocks, 256, 0, s>>>(p); else conv1d_gemm_kernel<2, 4><<<blocks64, 256, 0, s>>>(p); }
    else { if (small) conv1d_gemm_kernel<1, 2><<<blocks, 256, 0, s>>>(p); else conv1d_gemm_kernel<1, 8><<<blocks, 256, 0, s>>>(p); }
  } else {
    v3d::TimedScope ts(h->n_seg == 27 ? "sparse_conv_gemm" : h->n_seg == 3 ? "conv1d_gemm" : "linear_gemm", s);
#define V3D_GG(MBW_, NB_)                                                                  \
  do {                                                                                     \
    if (fp32_path) gemm_gather_kernel<MBW_, NB_, false><<<blocks, 256, 0, s>>>(p);         \
    else gemm_gather_kernel<MBW_, NB_, true><<<blocks, 256, 0, s>>>(p);                    \
  } while (0)
    // small M, split-bf16, every segment 16-byte loadable: the rounds kernel (four steps per pair of barriers)
    const int rounds_opt = v3d::option(v3d::kOptGemmRounds);
    bool rounds = (small || rounds_opt == 2) && !fp32_path && h->K % 4 == 0 && h->n_seg * (h->KP / kKC) >= 2 && h->KP / kKC <= 8 && rounds_opt != 0;
    for (int t = 0; rounds && t < h->n_seg; ++t)
      rounds = p.seg[t].ld % 4 == 0 && (reinterpret_cast<size_t>(p.seg[t].src) & 15) == 0;
    // a sparse convolution (one source, a row map per offset): the loader / matrix pipeline
    bool pipe = rounds && group_len == 0 && h->n_seg * (h->KP / kKC) >= 8 && (size_t)M * (size_t)p.seg[0].ld < ((size_t)1 << 31) &&
                v3d::option(v3d::kOptGemmPipe) != 0;
    for (int t = 0; pipe && t < h->n_seg; ++t)
      pipe = p.seg[t].idx != nullptr && p.seg[t].src == p.seg[0].src && p.seg[t].ld == p.seg[0].ld;
    if (pipe) {
      // rows per tile: a tile streams the whole weight image through its CU's vector-memory path (27 x K x N x 4 bytes), so
      // the tile is as tall as the number of workgroups allows -- about one per CU on the 128-channel levels, two on the wide
      // 64-channel level
      const int rows_opt = v3d::option(v3d::kOptGemmRoundRows);
      V3D_REQUIRE(rows_opt == 0 || rows_opt == 32 || rows_opt == 64 || rows_opt == 128, V3D_ERR_BAD_ARG,
                  "option gemm_round_rows must be 0, 32, 64 or 128 (got %d)", rows_opt);
      const int rows = rows_opt ? rows_opt : (h->MBW == 2 ? (M >= 8192 ? 64 : 32) : (M >= 8192 ? 64 : 32));
      const unsigned rb = (unsigned)((M + rows - 1) / rows);
      // measured on the cfg3 scene's levels (scripts/micro/sparse_ab.sh; rounds kernel -> here): 13 434 rows x 128 channels 84 -> 60 us
      // with 64-row tiles and 8 loader waves, 2 719 x 128: 58 -> 40 us with 32-row tiles and 4 loader waves, 59 975 x 64: 90 -> 84 us;
      // option gemm_pipe = 2 is the first version of the kernel (4 loader waves, fragments of 3 rounds in flight) for A/B runs
      const bool v2 = v3d::option(v3d::kOptGemmPipe) == 2;
#define V3D_PIPE(MBW_, NB_, NL_, MINB_)                                                                          \
  do {                                                                                                           \
    if (v2) gemm_gather_pipe_kernel<MBW_, NB_, 4, 4, 1><<<rb, 512, 0, s>>>(p);                                   \
    else gemm_gather_pipe_kernel<MBW_, NB_, NL_, 2, MINB_><<<rb, 256 + 64 * NL_, 0, s>>>(p);                     \
  } while (0)
      if (h->MBW == 2) {
        if (rows == 128) V3D_PIPE(2, 8, 8, 1);
        else if (rows == 64) V3D_PIPE(2, 4, 8, 1);
        else V3D_PIPE(2, 2, 4, 2);
      } else {
        if (rows == 128) V3D_PIPE(1, 8, 8, 1);
        else if (rows == 64) V3D_PIPE(1, 4, 8, 1);
        else V3D_PIPE(1, 2, 4, 2);
      }
#undef V3D_PIPE
    } else if (rounds) {
      // developer A/B: rows per tile (32 / 64 / 128) -- a tile re-reads the whole weight image, so L2 -> CU weight traffic is
      // M / rows x 27 x K x N x 4 bytes (0.8 GB per 64-channel conv on 60 k voxels with 32-row tiles, twice the gathers)
      const int rows_env = v3d::option(v3d::kOptGemmRoundRows);
      V3D_REQUIRE(rows_env == 0 || rows_env == 32 || rows_env == 64 || rows_env == 128, V3D_ERR_BAD_ARG,
                  "option gemm_round_rows must be 0, 32, 64 or 128 (got %d)", rows_env);
      const int rows = rows_env ? rows_env : small ? 32 : 128;
      const unsigned rb = (unsigned)((M + rows - 1) / rows);
      if (h->MBW == 2) {
        if (rows == 128) gemm_gather_rounds_kernel<2, 8, 2><<<rb, 256, 0, s>>>(p);
        else if (rows == 64) gemm_gather_rounds_kernel<2, 4, 4><<<rb, 256, 0, s>>>(p);
        else gemm_gather_rounds_kernel<2, 2, 4><<<rb, 256, 0, s>>>(p);
      } else {
        if (rows == 128) gemm_gather_rounds_kernel<1, 8, 2><<<rb, 256, 0, s>>>(p);
        else if (rows == 64) gemm_gather_rounds_kernel<1, 4, 4><<<rb, 256, 0, s>>>(p);
        else gemm_gather_rounds_kernel<1, 2, 4><<<rb, 256, 0, s>>>(p);
      }
    } else if (!small && v3d::option(v3d::kOptGemmRoundRows) == 64) {
      // developer A/B: 64-row tiles for large M on the one-step kernel (half the accumulators, one more workgroup per CU)
      const unsigned b64 = (unsigned)((M + 63) / 64);
#define V3D_GG64(MBW_)                                                                     \
  do {                                                                                     \
    if (fp32_path) gemm_gather_kernel<MBW_, 4, false><<<b64, 256, 0, s>>>(p);              \
    else gemm_gather_kernel<MBW_, 4, true><<<b64, 256, 0, s>>>(p);                         \
  } while (0)
      if (h->MBW == 2) V3D_GG64(2); else V3D_GG64(1);
#undef V3D_GG64
    } else if (h->MBW == 2) { if (small) V3D_GG(2, 2); else V3D_GG(2, 8); }
    else { if (small) V3D_GG(1, 2); else V3D_GG(1, 8); }
#undef V3D_GG
  }
  V3D_CHECK_LAUNCH("gemm_gather_kernel");
  return V3D_OK;
}

extern "C" int v3d_sparse_conv_f32(const v3d_gemm_weights* h, int M, const float* src, int ld_src, const int32_t* nbr,
                                   long long nbr_stride, int use_gn, float gn_eps, const float* residual, int ld_res,
                                   int relu_out, float* out, int ld_out, int precision, void* stream) {
  V3D_REQUIRE(h && src && nbr, V3D_ERR_BAD_ARG, "v3d_sparse_conv_f32: null argument");
  V3D_REQUIRE(nbr_stride >= M, V3D_ERR_BAD_SHAPE, "v3d_sparse_conv_f32: nbr_stride %lld < M %d", nbr_stride, M);
  const float* srcs[kMaxSeg];
  const int32_t* idxs[kMaxSeg];
  int lds[kMaxSeg];
  for (int s = 0; s < h->n_seg; ++s) {
    srcs[s] = src;
    idxs[s] = nbr + (size_t)s * (size_t)nbr_stride;
    lds[s] = ld_src;
  }
  return v3d_gemm_gather_f32(h, M, srcs, idxs, lds, 0, 0, use_gn, gn_eps, residual, ld_res, relu_out, nullptr, nullptr, 0, out,
                             ld_out, precision, stream);
}

extern "C" int v3d_fill_f32(float* ptr, size_t n, float value, void* stream) {
  V3D_REQUIRE(ptr || n == 0, V3D_ERR_BAD_ARG, "v3d_fill_f32: null pointer");
  if (n == 0) return V3D_OK;
  fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(ptr, n, value);
  V3D_CHECK_LAUNCH("fill_kernel");
  return V3D_OK;
}

// PointNet input rows (lightningmodel.py:180-183): x[i] = [pts[e1[i]] - anchor_pts[e0[i]] | pts_feat[e1[i]]] in one pass
// (the reference builds it with two index gathers, a subtraction, a third gather and a torch.cat).  4 threads per row.
namespace {
__global__ __launch_bounds__(256) void pointnet_input_kernel(const float* __restrict__ pts, const float* __restrict__ anchor,
                                                             const float* __restrict__ feat, const long long* __restrict__ e0,
                                                             const long long* __restrict__ e1, int n, int C, float* __restrict__ out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = gid >> 2;
  const int part = (int)(gid & 3);
  if (i >= n) return;
  const long long a = e0[i], q = e1[i];
  float* const o = out + (size_t)i * (3 + C);
  if (part == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = pts[(size_t)q * 3 + d] - anchor[(size_t)a * 3 + d];
  }
  const float* const f = feat + (size_t)q * C;
  for (int c = part; c < C; c += 4) o[3 + c] = f[c];
}
}  // namespace

extern "C" int v3d_pointnet_input_f32(const float* pts, const float* anchor_pts, const float* pts_feat, const int64_t* edge_anchor,
                                      const int64_t* edge_pt, int n_edges, int C, float* out, void* stream) {
  V3D_REQUIRE(pts && anchor_pts && pts_feat && edge_anchor && edge_pt && out, V3D_ERR_BAD_ARG, "v3d_pointnet_input_f32: null argument");
  V3D_REQUIRE(n_edges >= 0 && C >= 1, V3D_ERR_BAD_SHAPE, "v3d_pointnet_input_f32: bad shape");
  if (n_edges == 0) return V3D_OK;
  const long long threads = (long long)n_edges * 4;
  pointnet_input_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      pts, anchor_pts, pts_feat, (const long long*)edge_anchor, (const long long*)edge_pt, n_edges, C, out);
  V3D_CHECK_LAUNCH("pointnet_input_kernel");
  return V3D_OK;
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_gemm_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)8 * kPhaseSlots);
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gg_phase), h.size() * sizeof(unsigned long long)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < kPhaseSlots; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif
