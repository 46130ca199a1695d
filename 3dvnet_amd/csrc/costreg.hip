// Rows A5-A6 of SURVEY.md §8a: CostRegNet (dense 3D-conv U-Net, eval-mode BatchNorm folded) and the
// soft-argmin depth.  Reference semantics: mv3d/subnetworks/mvsnet.py:18-36,133-163,219-227.
//
// Two families of kernels live here.
//
// (1) The fused path (v3d_costreg_depth_f32 / _split): every layer on split-bf16 matrix cores.  Each fp32 operand
//     x = hi + lo (hi = RNE_bf16(x), lo = RNE_bf16(x - hi), 16 mantissa bits), each product hi*hi + hi*lo + lo*hi on
//     v_mfma_f32_16x16x32_bf16 with fp32 accumulation; activations travel between layers in the split channel-last
//     layout [n][C/8 groups][hi, lo][D][H][W][8 bf16] (4 bytes per value, every staging loop a 16-byte copy):
//       conv0_bf16x2_kernel      32 -> 8 at full resolution (68 % of the MACs), pair mode, 4 chunks of 8 channels
//       convg_bf16x2_kernel      conv1..conv6 (stride 1 / 2), 8 input channels per chunk, 16 output channels per workgroup
//       deconvg_bf16x2_kernel    conv7, conv8 (transposed conv as a GEMM over 2x2x2 output cells) + fp32 skip
//       conv9_prob_kernel        conv9 + conv0 skip + the 8 -> 1 prob conv, the 8-channel tensor stays in LDS
//       soft_argmin_kernel       softmax(-x) expectation over the depth planes
//     Final depth vs the fp32 CPU oracle: 5e-5 relative (gate 1e-4).
//
// (2) The exact-fp32 per-layer kernels (v3d_costreg_layer_f32, and the whole chain with precision = V3D_PRECISION_FP32):
//     one templated implicit-GEMM kernel on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain), described below, plus
//     prob_conv_kernel.  They were the round's first correct path and remain the reference point for the split
//     kernels' per-layer parity tests.
//
//   D[co, voxel] += W'[co, k] * X[k, voxel],  k = (input channel, kernel tap)
//
//   * A operand = BN-folded weights, pre-packed on the host into MFMA fragment order (one coalesced
//     256-B global load per fragment, L2-resident, shared by every workgroup);
//   * B operand = activations: a halo'd input tile of CK channels is staged in LDS channel-major
//     ([ck][z][y][x], plane stride chosen so the 4 k-lanes x 16 voxel-lanes of a ds_read_b32 hit
//     distinct banks); zero padding is materialised in the tile so the inner loop has no bounds
//     checks; every output voxel of the tile owns a per-lane LDS base offset, every kernel tap is a
//     wave-uniform offset on top of it;
//   * each wave keeps NBW voxel blocks x MB channel blocks of 16x16 accumulators in registers
//     across all input-channel chunks; epilogue = +bias, ReLU, optional skip add (after the ReLU,
//     mvsnet.py:159-161), store in the reference's [n, C, D, H, W] layout (16 consecutive voxels
//     per store row);
//   * stride-2 conv reads the tile with stride 2; the stride-2 transposed conv (k3, p1,
//     output_padding 1) is decomposed into its 8 output-parity classes, each a dense gather with
//     1..8 taps (out[o] = sum_k in[(o+1-k)/2] W[k] for (o+1-k) even).
//
// The unfused `prob` conv (8 -> 1 channel) is VALU work (a 1-wide GEMM would waste 15/16 of an MFMA)
// and the depth softmax + expectation is a per-pixel streaming reduction.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "v3d_common.h"

#ifndef V3D_ABLATE
#define V3D_ABLATE 0   // developer ablations of the conv kernel (1: no MFMA loop, 2: no restaging)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// kConvS1 / kConvS2: k3 p1 conv with stride 1 / 2.  kDeconvS2: k3 s2 p1 output_padding-1 transposed
// conv.  kConvS1Pair: stride-1 conv for COUT == 8 -- the 16 MFMA rows hold 2 x-shifts x 8 output
// channels over a 4-wide x window (36 "virtual taps", the weight image is zero where the shifted
// kernel does not reach), each voxel block is 16 x-PAIRS: 18 MFMAs per 16 output voxels instead of
// the 27 a half-empty 16-row tile would cost.
enum { kConvS1 = 0, kConvS2 = 1, kDeconvS2 = 2, kConvS1Pair = 3 };

#ifdef V3D_PHASE_TIMING
// developer build only (-DV3D_PHASE_TIMING): wave 0 of every workgroup accumulates the cycles between marks in
// registers and writes them to its own slot at the end (no atomics: they would stall the memory pipe being timed)
constexpr int kPhaseSlots = 1 << 16;
__device__ unsigned long long g_phase[8 * kPhaseSlots];
#define PHASE_DECL                                  \
  long long ph_t = __builtin_readcyclecounter();    \
  long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_MARK(i)                                   \
  do {                                                  \
    long long t_ = __builtin_readcyclecounter();        \
    ph_acc[i] += t_ - ph_t;                             \
    ph_t = t_;                                          \
  } while (0)
#define PHASE_FLUSH                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < kPhaseSlots)                                                 \
      for (int i_ = 0; i_ < 8; ++i_) g_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_];   \
  } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_FLUSH
#endif

struct ConvParams {
  const float* in;
  const float* wp;     // packed A fragments
  const float* bias;   // [COUT]
  const float* skip;   // [n, COUT, Do, Ho, Wo] or null
  float* out;
  int n, Di, Hi, Wi, Do, Ho, Wo, ntz, nty, ntx;
  int relu;
  int zy_order;        // tile order inside a view: 0 = x, y, z (z slowest), 1 = x, z, y (conv0 on large planes, see tile_order())
};

// The workgroups resident on an XCD (64 tiles of conv0 / conv9+prob) should be neighbours in the directions with the most
// halo: a 4 x 8 x 28 tile re-reads 50 % in z, 25 % in y, 7 % in x.  With x, y, z order they cover several whole z slabs when
// a slab is small (cfg2: 2 x 7 = 14 tiles) but only part of one when it is large (cfg5: 6 x 15 = 90 tiles: conv0 fetched
// 1.5x its input); then x, z, y order puts z neighbours side by side.
inline int tile_order(int ntx, int nty) {
#ifdef V3D_TILE_ORDER_XYZ      // developer A/B
  return 0;
#else
  return ntx * nty > 32 ? 1 : 0;
#endif
}

template <int MODE_, int CIN_, int COUT_, int TD_, int TH_, int TW_, int CK_, int OCC_ = 2>
struct ConvCfg {
  static constexpr int OCC = OCC_;   // min waves per SIMD (register budget 512 / OCC)
  static constexpr int MODE = MODE_, CIN = CIN_, COUT = COUT_, TD = TD_, TH = TH_, TW = TW_, CK = CK_;
  static constexpr bool S1LIKE = MODE == kConvS1 || MODE == kConvS1Pair;
  static constexpr int MB = MODE == kConvS1Pair ? 1 : (COUT + 15) / 16;
  static constexpr int ID = S1LIKE ? TD + 2 : MODE == kConvS2 ? 2 * TD + 1 : TD / 2 + 1;
  static constexpr int IH = S1LIKE ? TH + 2 : MODE == kConvS2 ? 2 * TH + 1 : TH / 2 + 1;
  static constexpr int IW = S1LIKE ? TW + 2 : MODE == kConvS2 ? 2 * TW + 1 : TW / 2 + 1;
  static constexpr int PLANE = ID * IH * IW;
  // channel-plane stride in LDS: odd where lanes read with x stride 2, == 16 (mod 32) otherwise
  static constexpr int S = (MODE == kConvS2 || MODE == kConvS1Pair)
                               ? (PLANE | 1) : ((PLANE - 16 + 31) / 32) * 32 + 16;
  static constexpr int KXN = MODE == kConvS1Pair ? 4 : 3;   // x taps
  static constexpr int NT = 9 * KXN;                        // (virtual) taps per input channel
  static constexpr int NVOX = TD * TH * TW;
  static constexpr int NCLS = MODE == kDeconvS2 ? 8 : 1;
  // lattice the voxel blocks enumerate: parity sub-lattice (deconv), x pairs (pair mode)
  static constexpr int CD = MODE == kDeconvS2 ? TD / 2 : TD;
  static constexpr int CH = MODE == kDeconvS2 ? TH / 2 : TH;
  static constexpr int CW = (MODE == kDeconvS2 || MODE == kConvS1Pair) ? TW / 2 : TW;
  static constexpr int NVC = CD * CH * CW;
  static constexpr int NBC = (NVC + 15) / 16;
  static constexpr int NBT = NBC * NCLS;
  static constexpr int NBW = (NBT + 3) / 4;
  static constexpr int NCHUNK = CIN / CK;
  static constexpr int C4 = CK / 4;
  static constexpr int LDS_BYTES = CK * S * 4;
  // staging geometry
  static constexpr int IWP = IW <= 8 ? 8 : IW <= 16 ? 16 : IW <= 32 ? 32 : 64;
  static constexpr int RPW = 64 / IWP;                     // rows per wave per iteration
  static constexpr int ROWS = CK * ID * IH;
  static constexpr int NIT = (ROWS + 4 * RPW - 1) / (4 * RPW);
  static constexpr int SU = NIT < 10 ? NIT : 10;           // rows in flight per lane
  // Prefetch mode: the next channel chunk's tile rows AND its weight fragments are loaded into
  // registers right after the barrier and stay in flight during the whole MFMA loop; A fragments are
  // then served from LDS (lgkmcnt) so that no vmcnt wait inside the loop drains the prefetch.
  static constexpr int WCH = NT * C4 * MB * 64;            // packed weight floats per chunk
  static constexpr int NWIT = (WCH + 255) / 256;
  static constexpr bool PF = WCH <= 8192 && (NIT + NWIT) <= 64 && (NIT + NWIT + 4 * NBW * MB) <= 150;
  static constexpr int TROWS = NIT * 4 * RPW;              // staging row-table entries (>= ROWS)
  static constexpr int LDS_FLOATS = CK * S + (PF ? WCH : 0) + 2 * TROWS;
  static_assert(IW <= 64, "tile row wider than a wave");
  static_assert(CIN % CK == 0 && CK % 4 == 0, "channel chunking");
  static_assert(MODE != kDeconvS2 || (TD % 2 == 0 && TH % 2 == 0 && TW % 2 == 0), "even tile");
  static_assert(MODE != kDeconvS2 || NBW == 2 * NBC, "two parity classes per wave");
  static_assert(MODE != kConvS1Pair || (COUT == 8 && TW % 2 == 0), "pair mode: 8 channels, even TW");
  static_assert(LDS_FLOATS * 4 <= 64 * 1024, "LDS tile");
};

// VEC (stride-1 layers in prefetch mode, tile and volume widths multiples of 4): the halo'd rows are fetched as aligned float4s
// (16 lanes per row, [ix0 - 3, ix0 + 61)) instead of one float per lane -- the CU's vector-memory path charges a 4-byte-per-lane
// wave instruction twice the cycles of a 16-byte one for a quarter of the bytes.
template <class C, bool VEC>
__global__ __launch_bounds__(256, C::OCC) void conv3d_mfma_kernel(ConvParams p) {
  constexpr int MODE = C::MODE;
  __shared__ float xs[C::LDS_FLOATS];
  float* const ws = xs + C::CK * C::S;   // weight fragments of the current chunk (PF mode)
  // per-row staging tables, built once: global element offset of the row start (kRowOob if the
  // row lies outside the volume in z/y) and LDS element offset of the row (-1 = no such row)
  int* const rowg = reinterpret_cast<int*>(xs + C::CK * C::S + (C::PF ? C::WCH : 0));
  int* const rowd = rowg + C::TROWS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;

  int b = v3d::xcd_contiguous_block();     // neighbouring tiles (shared halo) on the same XCD's L2
  const int tx = b % p.ntx; b /= p.ntx;
  const int ty = b % p.nty; b /= p.nty;
  const int tz = b % p.ntz;
  const int n = b / p.ntz;
  const int oz0 = tz * C::TD, oy0 = ty * C::TH, ox0 = tx * C::TW;
  const int iz0 = C::S1LIKE ? oz0 - 1 : MODE == kConvS2 ? 2 * oz0 - 1 : oz0 / 2;
  const int iy0 = C::S1LIKE ? oy0 - 1 : MODE == kConvS2 ? 2 * oy0 - 1 : oy0 / 2;
  const int ix0 = C::S1LIKE ? ox0 - 1 : MODE == kConvS2 ? 2 * ox0 - 1 : ox0 / 2;

  // per-lane LDS base offset of each of this wave's voxel blocks
  int boff[C::NBW];
#pragma unroll
  for (int j = 0; j < C::NBW; ++j) {
    int i = MODE == kDeconvS2 ? (j % C::NBC) : (wave * C::NBW + j);
    int v = min(i * 16 + jn, C::NVC - 1);
    int z = v / (C::CH * C::CW), y = (v / C::CW) % C::CH, x = v % C::CW;
    int off = MODE == kConvS2 ? ((2 * z) * C::IH + 2 * y) * C::IW + 2 * x
            : MODE == kConvS1Pair ? (z * C::IH + y) * C::IW + 2 * x
                                  : (z * C::IH + y) * C::IW + x;
    boff[j] = off + kq * C::S;
  }

  f32x4 acc[C::NBW][C::MB];
#pragma unroll
  for (int j = 0; j < C::NBW; ++j)
#pragma unroll
    for (int m = 0; m < C::MB; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const size_t in_plane = (size_t)p.Di * p.Hi * p.Wi;
  const float* inb = p.in + (size_t)n * C::CIN * in_plane;
  const float* wl = p.wp + lane;

  // A fragments (weights) are software-pipelined one tap ahead: fragment index f = chunk*NT + tap
  // walks the packed image linearly, so "next tap" also crosses chunk boundaries.
  constexpr int kLastFrag = C::NCHUNK * C::NT - 1;
  float a_cur[C::C4][C::MB], a_nxt[C::C4][C::MB];
  auto load_a = [&](float (&a)[C::C4][C::MB], int f) {
#pragma unroll
    for (int c4 = 0; c4 < C::C4; ++c4)
#pragma unroll
      for (int m = 0; m < C::MB; ++m) a[c4][m] = wl[((size_t)(f * C::C4 + c4) * C::MB + m) * 64];
  };
  if constexpr (MODE != kDeconvS2 && !C::PF) load_a(a_cur, 0);

  // staging lane role: a group of IWP lanes (IWP = IW rounded up to a power of two) copies one
  // (channel, z, y) row of the halo'd tile
  const int lrow = lane / C::IWP, lx = lane % C::IWP;
  const int sgx = ix0 + lx;
  const bool xok = lx < C::IW;
  const bool xin = xok && sgx >= 0 && sgx < p.Wi;
  constexpr int kRowOob = -2147483647 - 1;
  for (int r = tid; r < C::TROWS; r += 256) {
    const int ck = r / (C::ID * C::IH), rz = (r / C::IH) % C::ID, ry = r % C::IH;
    const int gz = iz0 + rz, gy = iy0 + ry;
    const bool in_vol = gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
    rowg[r] = (r < C::ROWS && in_vol) ? ck * (int)in_plane + (gz * p.Hi + gy) * p.Wi + ix0 : kRowOob;
    rowd[r] = r < C::ROWS ? ck * C::S + (rz * C::IH + ry) * C::IW : -1;
  }
  __syncthreads();
  const int myrow0 = wave * C::RPW + lrow;        // this lane's rows: myrow0 + it * 4 * RPW

  constexpr int NITV = (C::ROWS + 15) / 16;     // VEC: 16 rows per workgroup iteration
  const int vj = tid & 15, vrow = tid >> 4;
  const int vgx = ix0 - 3 + 4 * vj;
  const bool vin = vgx >= 0 && vgx + 3 < p.Wi && 4 * vj - 3 < C::IW;
  float pre[(C::PF && !VEC) ? C::NIT : 1], wreg[C::PF ? C::NWIT : 1];
  f32x4 pre4[(C::PF && VEC) ? NITV : 1];
  auto pf_issue = [&](int chunk) {
    if constexpr (VEC) {
      const float* incv = inb + (size_t)chunk * C::CK * in_plane + (4 * vj - 3);
#pragma unroll
      for (int it = 0; it < NITV; ++it) {
        const int r = it * 16 + vrow;
        const int g = rowg[min(r, C::TROWS - 1)];
        pre4[it] = (r < C::ROWS && g != kRowOob && vin) ? *reinterpret_cast<const f32x4*>(incv + g) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    } else {
      const float* inc = inb + (size_t)chunk * C::CK * in_plane + lx;
#pragma unroll
      for (int it = 0; it < C::NIT; ++it) {
        const int g = rowg[myrow0 + it * 4 * C::RPW];
        pre[it] = (g != kRowOob && xin) ? inc[g] : 0.f;
      }
    }
    const float* wc = p.wp + (size_t)chunk * C::WCH + tid;
#pragma unroll
    for (int i = 0; i < C::NWIT; ++i) wreg[i] = (i * 256 + tid < C::WCH) ? wc[i * 256] : 0.f;
  };
  auto pf_commit = [&]() {
    if constexpr (VEC) {
#pragma unroll
      for (int it = 0; it < NITV; ++it) {
        const int r = it * 16 + vrow;
        const int d = rowd[min(r, C::TROWS - 1)];
        if (r < C::ROWS && d >= 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int x = 4 * vj - 3 + e;
            if (x >= 0 && x < C::IW) xs[d + x] = pre4[it][e];
          }
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < C::NIT; ++it) {
        const int d = rowd[myrow0 + it * 4 * C::RPW];
        if (d >= 0 && xok) xs[d + lx] = pre[it];
      }
    }
#pragma unroll
    for (int i = 0; i < C::NWIT; ++i)
      if (i * 256 + tid < C::WCH) ws[i * 256 + tid] = wreg[i];
  };
  PHASE_DECL;
  if constexpr (C::PF) pf_issue(0);
  PHASE_MARK(0);

#pragma unroll 1
  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    __syncthreads();
    PHASE_MARK(1);
    if constexpr (C::PF) {
#if V3D_ABLATE == 2
      if (chunk == 0) pf_commit();
      __syncthreads();
#else
      pf_commit();
      PHASE_MARK(2);
      __syncthreads();
      PHASE_MARK(3);
      if (chunk + 1 < C::NCHUNK) pf_issue(chunk + 1);
      PHASE_MARK(4);
#endif
    } else {
      // batched staging: SU independent rows are in flight per lane before the LDS writes, so
      // global latency is paid once per batch rather than once per element
      const float* inc = inb + (size_t)chunk * C::CK * in_plane + lx;
#pragma unroll 1
      for (int it0 = 0; it0 < C::NIT; it0 += C::SU) {
        float v[C::SU];
#pragma unroll
        for (int u = 0; u < C::SU; ++u) {
          v[u] = 0.f;
          if (it0 + u < C::NIT) {
            const int g = rowg[myrow0 + (it0 + u) * 4 * C::RPW];
            if (g != kRowOob && xin) v[u] = inc[g];
          }
        }
#pragma unroll
        for (int u = 0; u < C::SU; ++u) {
          if (it0 + u < C::NIT) {
            const int d = rowd[myrow0 + (it0 + u) * 4 * C::RPW];
            if (d >= 0 && xok) xs[d + lx] = v[u];
          }
        }
      }
      __syncthreads();
    }

#if V3D_ABLATE == 1
    if (p.relu == 12345)
#endif
    if constexpr (MODE != kDeconvS2) {
#pragma unroll 1
      for (int kzy = 0; kzy < 9; ++kzy) {
        const int kz = kzy / 3, ky = kzy % 3;
#pragma unroll
        for (int kx = 0; kx < C::KXN; ++kx) {
          const int tap = kzy * C::KXN + kx;
          const int tapoff = (kz * C::IH + ky) * C::IW + kx;
          if constexpr (C::PF) {
#pragma unroll
            for (int c4 = 0; c4 < C::C4; ++c4)
#pragma unroll
              for (int m = 0; m < C::MB; ++m)
                a_cur[c4][m] = ws[((tap * C::C4 + c4) * C::MB + m) * 64 + lane];
          } else {
            load_a(a_nxt, min(chunk * C::NT + tap + 1, kLastFrag));
          }
#pragma unroll
          for (int c4 = 0; c4 < C::C4; ++c4) {
#pragma unroll
            for (int j = 0; j < C::NBW; ++j) {
              const float bv = xs[boff[j] + tapoff + c4 * 4 * C::S];
#pragma unroll
              for (int m = 0; m < C::MB; ++m)
                acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[c4][m], bv, acc[j][m], 0, 0, 0);
            }
          }
          if constexpr (!C::PF) {
#pragma unroll
            for (int c4 = 0; c4 < C::C4; ++c4)
#pragma unroll
              for (int m = 0; m < C::MB; ++m) a_cur[c4][m] = a_nxt[c4][m];
          }
        }
      }
    } else {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        const int cls = wave * 2 + cl;
        const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll 1
        for (int tap = 0; tap < 27; ++tap) {
          const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
          // parity p == 0 (even output): only k = 1; p == 1: k = 0 (input c+1) and k = 2 (input c)
          if ((pz == 0) != (kz == 1)) continue;
          if ((py == 0) != (ky == 1)) continue;
          if ((px == 0) != (kx == 1)) continue;
          const int dz = kz == 0 ? 1 : 0, dy = ky == 0 ? 1 : 0, dx = kx == 0 ? 1 : 0;
          const int tapoff = (dz * C::IH + dy) * C::IW + dx;
          if constexpr (C::PF) {
#pragma unroll
            for (int c4 = 0; c4 < C::C4; ++c4)
#pragma unroll
              for (int m = 0; m < C::MB; ++m)
                a_cur[c4][m] = ws[((tap * C::C4 + c4) * C::MB + m) * 64 + lane];
          } else {
            load_a(a_cur, chunk * 27 + tap);
          }
#pragma unroll
          for (int c4 = 0; c4 < C::C4; ++c4) {
#pragma unroll
            for (int i = 0; i < C::NBC; ++i) {
              const float bv = xs[boff[cl * C::NBC + i] + tapoff + c4 * 4 * C::S];
#pragma unroll
              for (int m = 0; m < C::MB; ++m)
                acc[cl * C::NBC + i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    a_cur[c4][m], bv, acc[cl * C::NBC + i][m], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  PHASE_MARK(5);
  // ---- epilogue: bias, ReLU, skip, store ([n, COUT, Do, Ho, Wo]) -------------------------------
  const size_t out_plane = (size_t)p.Do * p.Ho * p.Wo;
  // Where the whole output tile fits the (dead) input tile's LDS and rows are 16-byte addressable, it leaves through LDS as
  // float4 row segments with float4 skip loads.  The direct paths below issue 4- or 8-byte skip loads and stores per (channel,
  // voxel) and lane: on conv9 (8 channels at full resolution) that epilogue was 75 % of the workgroup's time.
  constexpr int OCS = C::NVOX + 4;                 // channel stride of the staged tile (padded against bank conflicts)
  constexpr bool kStage = C::COUT * OCS <= C::LDS_FLOATS && C::TW % 4 == 0;
  const bool staged = kStage && (p.Wo & 3) == 0 &&
                      ((reinterpret_cast<size_t>(p.out) | reinterpret_cast<size_t>(p.skip)) & 15) == 0;
  float* const os = xs;
  auto stage_value = [&](int co, int z, int y, int x, float val) __attribute__((always_inline)) {
    val += p.bias[co];
    if (p.relu) val = fmaxf(val, 0.f);
    os[co * OCS + (z * C::TH + y) * C::TW + x] = val;
  };
  auto store_staged = [&]() __attribute__((always_inline)) {
    __syncthreads();
    constexpr int QPR = C::TW / 4, NQ = C::COUT * C::TD * C::TH * QPR;      // float4 per row, per tile
#pragma unroll 4
    for (int q4 = tid; q4 < NQ; q4 += 256) {
      const int co = q4 / (C::TD * C::TH * QPR), rem = q4 % (C::TD * C::TH * QPR);
      const int row = rem / QPR, xq4 = rem % QPR;
      const int gz = oz0 + row / C::TH, gy = oy0 + row % C::TH, gx = ox0 + 4 * xq4;
      if (gz >= p.Do || gy >= p.Ho || gx >= p.Wo) continue;     // Wo % 4 == 0: a float4 is inside or outside as a whole
      f32x4 val = *reinterpret_cast<const f32x4*>(os + co * OCS + row * C::TW + 4 * xq4);
      const size_t o = ((size_t)n * C::COUT + co) * out_plane + ((size_t)gz * p.Ho + gy) * p.Wo + gx;
      if (p.skip) val += *reinterpret_cast<const f32x4*>(p.skip + o);
      *reinterpret_cast<f32x4*>(p.out + o) = val;
    }
  };
  if (staged) __syncthreads();                     // every wave is done reading the input tile / weight fragments
  if constexpr (MODE == kDeconvS2) {
    // This wave owns output parities (pz, py) = (wave >> 1, wave & 1) and BOTH x parities
    // (accumulator halves cl = 0 / 1), so lane jn holds x = 2 xc and 2 xc + 1: one float2 store per
    // (channel, voxel pair) -> 16 lanes write 128 contiguous bytes.
    const int pz = (wave >> 1) & 1, py = wave & 1;
    if (staged) {
#pragma unroll
      for (int i = 0; i < C::NBC; ++i) {
        const int v = i * 16 + jn;
        if (v >= C::NVC) continue;
        const int z = 2 * (v / (C::CH * C::CW)) + pz, y = 2 * ((v / C::CW) % C::CH) + py, x = 2 * (v % C::CW);
#pragma unroll
        for (int m = 0; m < C::MB; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = m * 16 + kq * 4 + r;
            if (co < C::COUT) { stage_value(co, z, y, x, acc[i][m][r]); stage_value(co, z, y, x + 1, acc[C::NBC + i][m][r]); }
          }
      }
      store_staged();
      PHASE_MARK(6);
      PHASE_FLUSH;
      return;
    }
#pragma unroll
    for (int i = 0; i < C::NBC; ++i) {
      const int v = i * 16 + jn;
      if (v >= C::NVC) continue;
      const int z = 2 * (v / (C::CH * C::CW)) + pz, y = 2 * ((v / C::CW) % C::CH) + py;
      const int x = 2 * (v % C::CW);
      const int gz = oz0 + z, gy = oy0 + y, gx = ox0 + x;
      if (gz >= p.Do || gy >= p.Ho || gx >= p.Wo) continue;      // Wo is even: gx + 1 < Wo too
      const size_t sp = ((size_t)gz * p.Ho + gy) * p.Wo + gx;
#pragma unroll
      for (int m = 0; m < C::MB; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = m * 16 + kq * 4 + r;
          if (co < C::COUT) {
            const float bsv = p.bias[co];
            float2 val = make_float2(acc[i][m][r] + bsv, acc[C::NBC + i][m][r] + bsv);
            if (p.relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); }
            const size_t o = ((size_t)n * C::COUT + co) * out_plane + sp;
            if (p.skip) {
              const float2 sk = *reinterpret_cast<const float2*>(p.skip + o);
              val.x += sk.x; val.y += sk.y;
            }
            *reinterpret_cast<float2*>(p.out + o) = val;
          }
        }
      }
    }
  } else if constexpr (MODE == kConvS1Pair) {
    // rows 0-7: x shift 0, rows 8-15: x shift 1 -> lane quarter kq holds channels 4*(kq&1)+r at
    // x = 2*pair + (kq >> 1); quarters kq and kq+2 interleave into full 128-B runs per channel.
    const int sx = kq >> 1, cbase = 4 * (kq & 1);
    if (staged) {
#pragma unroll
      for (int j = 0; j < C::NBW; ++j) {
        const int v = (wave * C::NBW + j) * 16 + jn;
        if (v >= C::NVC) continue;
        const int z = v / (C::CH * C::CW), y = (v / C::CW) % C::CH, x = 2 * (v % C::CW) + sx;
#pragma unroll
        for (int r = 0; r < 4; ++r) stage_value(cbase + r, z, y, x, acc[j][0][r]);
      }
      store_staged();
      PHASE_MARK(6);
      PHASE_FLUSH;
      return;
    }
#pragma unroll
    for (int j = 0; j < C::NBW; ++j) {
      const int v = (wave * C::NBW + j) * 16 + jn;
      if (v >= C::NVC) continue;
      const int z = v / (C::CH * C::CW), y = (v / C::CW) % C::CH, x = 2 * (v % C::CW) + sx;
      const int gz = oz0 + z, gy = oy0 + y, gx = ox0 + x;
      if (gz >= p.Do || gy >= p.Ho || gx >= p.Wo) continue;
      const size_t sp = ((size_t)gz * p.Ho + gy) * p.Wo + gx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cbase + r;
        float val = acc[j][0][r] + p.bias[co];
        if (p.relu) val = fmaxf(val, 0.f);
        const size_t o = ((size_t)n * C::COUT + co) * out_plane + sp;
        if (p.skip) val += p.skip[o];
        p.out[o] = val;
      }
    }
  } else {
    if (staged) {
#pragma unroll
      for (int j = 0; j < C::NBW; ++j) {
        const int v = (wave * C::NBW + j) * 16 + jn;
        if (v >= C::NVC) continue;
        const int z = v / (C::CH * C::CW), y = (v / C::CW) % C::CH, x = v % C::CW;
#pragma unroll
        for (int m = 0; m < C::MB; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = m * 16 + kq * 4 + r;
            if (co < C::COUT) stage_value(co, z, y, x, acc[j][m][r]);
          }
      }
      store_staged();
      PHASE_MARK(6);
      PHASE_FLUSH;
      return;
    }
#pragma unroll
    for (int j = 0; j < C::NBW; ++j) {
      const int v = (wave * C::NBW + j) * 16 + jn;
      if (v >= C::NVC) continue;
      const int z = v / (C::CH * C::CW), y = (v / C::CW) % C::CH, x = v % C::CW;
      const int gz = oz0 + z, gy = oy0 + y, gx = ox0 + x;
      if (gz >= p.Do || gy >= p.Ho || gx >= p.Wo) continue;
      const size_t sp = ((size_t)gz * p.Ho + gy) * p.Wo + gx;
#pragma unroll
      for (int m = 0; m < C::MB; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = m * 16 + kq * 4 + r;
          if (co < C::COUT) {
            float val = acc[j][m][r] + p.bias[co];
            if (p.relu) val = fmaxf(val, 0.f);
            const size_t o = ((size_t)n * C::COUT + co) * out_plane + sp;
            if (p.skip) val += p.skip[o];
            p.out[o] = val;
          }
        }
      }
    }
  }
  PHASE_MARK(6);
  PHASE_FLUSH;
}

// ---- conv0 on split-bf16 matrix cores -----------------------------------------------------------------
// conv0 (32 -> 8 channels at full resolution) carries 68 % of the regulariser's MACs.  bf16 MFMA runs at 16x
// the fp32-MFMA rate, so every fp32 operand is split x = hi + lo into two bf16 values (hi = RNE(x),
// lo = RNE(x - hi): 16 mantissa bits together) and the product is evaluated as hi*hi + hi*lo + lo*hi with
// fp32 accumulation: 3 bf16 MFMAs (K = 32) replace 8 fp32 MFMAs (K = 4).  Measured error of this layer
// 7e-6 of max|out| (fp32 MFMA: 7e-7) and <= 1e-5 relative on the final depth (gate: 1e-4) -- DESIGN.md.
// Same pair-mode geometry as the fp32 kernel: rows = 2 x-shifts x 8 output channels; one MFMA covers the 4 x
// taps of a (kz, ky) kernel row (k = 8*kx' + ci) for a group of 8 input channels.  LDS holds the halo'd tile
// channel-last in 16-byte slots (8 bf16 of one voxel), one array for hi and one for lo, so a B fragment is
// a single ds_read_b128; A fragments (weights, split on the host) sit in LDS per channel group.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct C0 {
  static constexpr int TD = 4, TH = 8, TW = 28, ID = TD + 2, IH = TH + 2, IW = TW + 2;
  static constexpr int NVOXI = ID * IH * IW;                 // 1800 voxels in the halo'd tile
  static constexpr int CG = 8, NCH = 32 / CG;                // channels per chunk, chunks
  static constexpr int SROWS = ID * IH;                      // 60 spatial rows (z, y) of IW voxels
  static constexpr int NITS = (SROWS + 7) / 8;               // 8 lane groups of 32 lanes per iteration
  static constexpr int WU32 = 9 * 2 * 64 * 4;                // weight words per chunk (hi + lo fragments)
  static constexpr int NWIT = WU32 / 256;                    // 18
  static constexpr size_t LDS_BYTES = (size_t)NVOXI * 16 * 2 + (size_t)WU32 * 4 + 3 * 64 * 4;
  static_assert(WU32 % 256 == 0, "geometry");
};

// cond ? a : b as a bit select (v_bfi_b32): written with ?: on two arrays the compiler turns the pair into a private array
// indexed by the lane, i.e. scratch memory and a vector load the epilogue then has to wait for.
__device__ __forceinline__ float lane_select(int cond, float a, float b) {
  const unsigned m = 0u - (unsigned)(cond != 0);
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

__device__ __forceinline__ unsigned bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// The same split on packed pairs (v_cvt_pk_bf16_f32: round to nearest even in hardware, the bits of bf16_rne for finite values;
// psv_variance.hip / conv0z.hip / conv12z.hip use it too): 12 instead of ~50 vector instructions per four values in the epilogues
__device__ __forceinline__ unsigned cr_pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, u32x2& hp, u32x2& lp) {
  hp = (u32x2){cr_pack_bf16x2(a, b), cr_pack_bf16x2(c, d)};
  lp = (u32x2){cr_pack_bf16x2(a - __uint_as_float(hp.x << 16), b - __uint_as_float(hp.x & 0xffff0000u)),
               cr_pack_bf16x2(c - __uint_as_float(hp.y << 16), d - __uint_as_float(hp.y & 0xffff0000u))};
}

// SPLIT_IN: `p.in` is the split-bf16 volume written by psv_variance_kernel<32, true>
// ([n][4 chunks][hi, lo][D][H][W] 16-byte slots): staging is then 16-byte copies, no conversion.
// SPLIT_OUT: the 8 output channels of a voxel leave as one hi slot and one lo slot of the split channel-last layout
// ([n][hi, lo][D][H][W] 16-byte slots) consumed by convh_bf16x2_kernel (conv1) and conv9_prob_kernel (skip).
// NW = waves per workgroup (4 or 8): with 8, a wave owns one output row of the 4 planes and a CU holds 4 waves per SIMD
// (2 workgroups), so one wave's LDS / barrier waits are covered by another's MFMAs.
// conv0's input loads are plain loads: with the non-temporal hint the halo rows a neighbouring tile reads a moment later
// were not kept in L2 (-DV3D_C0_NT_LOAD: 1.135 vs 1.09 ms per 64 views, two alternating builds on one box).
#ifdef V3D_C0_NT_LOAD
#define V3D_C0_LOAD(p) __builtin_nontemporal_load(p)
#else
#define V3D_C0_LOAD(p) (*(p))
#endif
#ifndef V3D_C0_ABLATE
#define V3D_C0_ABLATE 0      // developer ablations (scripts/ab_build.sh): 1 no MFMAs (0.82 ms), 2 no input loads (0.84), 3 no LDS commit (0.97), 4 no output stores (1.06), 5 stores into a 2 MB window (1.07), 6 loads from a 1 MB window (1.02); full kernel 1.09 ms per 64 views
#endif
// The kernel is written as a tile walk (v3d::xcd_tile_walk): with a grid of the workgroups the chip holds at once
// (-DV3D_C0_PERSIST) a workgroup walks tiles first + i, first + i + G, ... of its XCD's contiguous run and the chunk pipeline
// runs straight across tile boundaries (the first chunk of the next tile is requested during the last MFMA phase of the
// current one).  Measured equal to one tile per workgroup (1.09 vs 1.11 ms per 64 views) with 15 % more halo traffic
// (4.30 vs 3.74 GB fetched: the resident workgroups run in lock step and share less in L2), so the default grid is one
// workgroup per tile: the kernel is bound by the per-chunk chain load -> commit -> barrier -> MFMA with one chunk of
// prefetch (registers) and a single LDS buffer (ablations at V3D_C0_ABLATE), not by tile turnover.
template <bool SPLIT_IN, bool SPLIT_OUT, int NW>
__global__ __launch_bounds__(64 * NW, 2 * NW / 4) void conv0_bf16x2_kernel(ConvParams p) {
  constexpr int NT = 64 * NW, RPI = NT / 32, RPW = C0::TH / NW;      // threads, staging rows per iteration, rows per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* const xh = reinterpret_cast<u32x4*>(smem);                           // [NVOXI] hi slots
  u32x4* const xl = xh + C0::NVOXI;                                           // [NVOXI] lo slots
  unsigned* const wsu = reinterpret_cast<unsigned*>(xl + C0::NVOXI);          // [WU32] weight fragments
  int* const rowd = reinterpret_cast<int*>(wsu + C0::WU32);                   // [64] first voxel of a spatial row in the tile
  int* const rowg2 = rowd + 64;                                               // [2][64] global offset of the row, per tile parity

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  const size_t in_plane = (size_t)p.Di * p.Hi * p.Wi;
  constexpr int kRowOob = -2147483647 - 1;

  struct Tile { int n, oz0, oy0, ox0; };
  auto decode = [&](int t) __attribute__((always_inline)) {
    Tile q;
    const int tx = t % p.ntx; t /= p.ntx;
    int ty, tz;
    if (p.zy_order) { tz = t % p.ntz; t /= p.ntz; ty = t % p.nty; q.n = t / p.nty; }
    else { ty = t % p.nty; t /= p.nty; tz = t % p.ntz; q.n = t / p.ntz; }
    q.oz0 = tz * C0::TD; q.oy0 = ty * C0::TH; q.ox0 = tx * C0::TW;
    return q;
  };
  auto fill_rows = [&](const Tile& q, int par) __attribute__((always_inline)) {
    if (tid < 64) {
      const int rz = tid / C0::IH, ry = tid % C0::IH;
      const int gz = q.oz0 - 1 + rz, gy = q.oy0 - 1 + ry;
      const bool ok = tid < C0::SROWS && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
      rowg2[par * 64 + tid] = ok ? (gz * p.Hi + gy) * p.Wi + q.ox0 - 1 : kRowOob;
    }
  };
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n * p.ntz * p.nty * p.ntx);   // neighbouring tiles (shared halo) on one XCD's L2
  if (walk.t >= walk.end) return;
  Tile cur = decode(walk.t);
  if (tid < 64) {
    const int rz = tid / C0::IH, ry = tid % C0::IH;
    rowd[tid] = tid < C0::SROWS ? (rz * C0::IH + ry) * C0::IW : -1;
  }
  fill_rows(cur, 0);
  __syncthreads();

  // MFMA role: wave w owns output rows y = 2w, 2w+1 for all TD planes; one column block = the 14 x pairs of a row
  // (lanes 14, 15 idle), so an input row fetched from LDS feeds up to 3 output planes (z reuse).
  static_assert(C0::TH == 8 && C0::TW == 28, "wave -> row mapping");
  constexpr int NACC = C0::TD * RPW;
  f32x4 acc[NACC];

  // staging role: 32 lanes = x of one spatial row, 8 rows per iteration, 8 channels per lane
  const int grp = tid >> 5, lx = tid & 31;
  const bool xok = lx < C0::IW;
  constexpr int NITS_S = (2 * C0::SROWS + RPI - 1) / RPI;       // split input: 60 hi rows then 60 lo rows
  constexpr int NITS_F = (C0::SROWS + RPI - 1) / RPI;           // fp32 input: 60 rows x 8 channels
  constexpr int NWQ = (C0::WU32 / 4 + NT - 1) / NT;             // 16-byte weight loads per thread
  float pre[SPLIT_IN ? 1 : NITS_F][C0::CG];
  u32x4 pres[SPLIT_IN ? NITS_S : 1];
  u32x4 wreg[NWQ];
  // The loads of the next chunk are issued in three parts between the MFMA groups of the current chunk: the
  // vector-memory pipe takes ~16 cycles per 1 KB instruction, which would otherwise stall the wave in front of
  // its MFMAs for the whole batch (measured 3.3k cycles per chunk).
  auto issue_part = [&](const Tile& q, int par, int chunk, auto part_c) __attribute__((always_inline)) {
    constexpr int part = decltype(part_c)::value;
    const int* const rowg = rowg2 + par * 64;
    const int sgx = q.ox0 - 1 + lx;
    const bool xin = xok && sgx >= 0 && sgx < p.Wi;
    if constexpr (SPLIT_IN) {
      const u32x4* const ins = reinterpret_cast<const u32x4*>(p.in) + (size_t)q.n * 8 * in_plane + lx;
      constexpr int per = (NITS_S + 2) / 3;
#pragma unroll
      for (int it = part * per; it < (part + 1) * per && it < NITS_S; ++it) {
        const int rr = it * RPI + grp;
        const int hl = rr >= C0::SROWS ? 1 : 0;
        const int g = rowg[rr - hl * C0::SROWS];               // rows 60..63 of the table are out of range
        pres[it] = (rr < 2 * C0::SROWS && g != kRowOob && xin && V3D_C0_ABLATE != 2)
                       ? V3D_C0_LOAD(ins + (V3D_C0_ABLATE == 6 ? (size_t)(g & 0xffff) : (size_t)(chunk * 2 + hl) * in_plane + g))
                       : (u32x4){0u, 0u, 0u, 0u};
      }
    } else {
      constexpr int per = (NITS_F + 2) / 3;
      const float* inc = p.in + ((size_t)q.n * 32 + (size_t)chunk * C0::CG) * in_plane + lx;
#pragma unroll
      for (int it = part * per; it < (part + 1) * per && it < NITS_F; ++it) {
        const int g = rowg[min(it * RPI + grp, 63)];
#pragma unroll
        for (int c = 0; c < C0::CG; ++c) pre[it][c] = (g != kRowOob && xin) ? inc[(size_t)c * in_plane + g] : 0.f;
      }
    }
    const u32x4* wc = reinterpret_cast<const u32x4*>(p.wp) + (size_t)chunk * (C0::WU32 / 4) + tid;
    constexpr int wper = (NWQ + 2) / 3;
#pragma unroll
    for (int i = part * wper; i < (part + 1) * wper && i < NWQ; ++i)
      wreg[i] = (i * NT + tid < C0::WU32 / 4) ? wc[i * NT] : (u32x4){0u, 0u, 0u, 0u};
  };
  auto commit = [&]() __attribute__((always_inline)) {
    if constexpr (SPLIT_IN) {
#pragma unroll
      for (int it = 0; it < NITS_S; ++it) {
        const int rr = it * RPI + grp;
        const int hl = rr >= C0::SROWS ? 1 : 0;
        if (rr < 2 * C0::SROWS && xok && (V3D_C0_ABLATE != 3 || pres[it][0] == 0x12345u)) (hl ? xl : xh)[rowd[rr - hl * C0::SROWS] + lx] = pres[it];
      }
    } else {
#pragma unroll
      for (int it = 0; it < NITS_F; ++it) {
        const int d = rowd[min(it * RPI + grp, 63)];
        if (d >= 0 && xok) {
          unsigned h[8], l[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            h[c] = bf16_rne(pre[it][c]);
            l[c] = bf16_rne(pre[it][c] - __uint_as_float(h[c] << 16));
          }
          xh[d + lx] = (u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
          xl[d + lx] = (u32x4){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NWQ; ++i)
      if (i * NT + tid < C0::WU32 / 4) reinterpret_cast<u32x4*>(wsu)[i * NT + tid] = wreg[i];
  };
  // one ky slice of the 27 taps: the 3 kz weight fragments stay in registers, every input row read from LDS feeds
  // up to 3 output planes
  auto mfma_ky = [&](int ky) __attribute__((always_inline)) {
    const u32x4* wf = reinterpret_cast<const u32x4*>(wsu) + lane;
    bf16x8 a_hi[3], a_lo[3];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      a_hi[kz] = __builtin_bit_cast(bf16x8, wf[((kz * 3 + ky) * 2) * 64]);
      a_lo[kz] = __builtin_bit_cast(bf16x8, wf[((kz * 3 + ky) * 2 + 1) * 64]);
    }
#pragma unroll
    for (int yy = 0; yy < RPW; ++yy) {
      const int rowbase = (wave * RPW + yy + ky) * C0::IW + 2 * jn + kq;
#pragma unroll
      for (int iz = 0; iz < C0::ID; ++iz) {
        const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xh[rowbase + iz * C0::IH * C0::IW]);
        const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xl[rowbase + iz * C0::IH * C0::IW]);
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          const int z = iz - kz;
          if (z >= 0 && z < C0::TD) acc[z * RPW + yy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[kz], b_hi, acc[z * RPW + yy], 0, 0, 0);
        }
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          const int z = iz - kz;
          if (z >= 0 && z < C0::TD) acc[z * RPW + yy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[kz], b_lo, acc[z * RPW + yy], 0, 0, 0);
        }
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          const int z = iz - kz;
          if (z >= 0 && z < C0::TD) acc[z * RPW + yy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[kz], b_hi, acc[z * RPW + yy], 0, 0, 0);
        }
      }
    }
  };

  // The bias reaches the epilogue through SGPRs: a per-lane vector load there makes the compiler wait for vmcnt(0) in front
  // of every conditionally stored accumulator, i.e. for the next tile's prefetch and for the stores just issued.
  float sbias[8];                          // wave-uniform addresses: scalar loads, eight SGPRs for the whole kernel
#pragma unroll
  for (int r = 0; r < 8; ++r) sbias[r] = p.bias[r];

  PHASE_DECL;
  issue_part(cur, 0, 0, std::integral_constant<int, 0>{});
  issue_part(cur, 0, 0, std::integral_constant<int, 1>{});
  issue_part(cur, 0, 0, std::integral_constant<int, 2>{});
  PHASE_MARK(0);
  int par = 0;
#pragma unroll 1
  for (int t = walk.t; t < walk.end; t += walk.step, par ^= 1) {
    const int tn = t + walk.step;
    const bool has_next = tn < walk.end;
    const Tile nxt = decode(has_next ? tn : t);
    // the row table of the next tile: written now, first read in the MFMA phase of this tile's last chunk (several barriers
    // later); its previous content (tile t - step) was last read in tile t - step's chunk 2
    if (has_next) fill_rows(nxt, par ^ 1);
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int chunk = 0; chunk < C0::NCH; ++chunk) {
      __syncthreads();
      PHASE_MARK(1);
      commit();
      PHASE_MARK(2);
      __syncthreads();
      PHASE_MARK(3);
      const bool last = chunk + 1 == C0::NCH;
      const bool more = !last || has_next;
      const Tile& lt = last ? nxt : cur;                  // tile / table / chunk the next staging round belongs to
      const int lpar = last ? par ^ 1 : par, lchunk = last ? 0 : chunk + 1;
      if (more) issue_part(lt, lpar, lchunk, std::integral_constant<int, 0>{});
      if (V3D_C0_ABLATE != 1) mfma_ky(0);
      if (more) issue_part(lt, lpar, lchunk, std::integral_constant<int, 1>{});
      if (V3D_C0_ABLATE != 1) mfma_ky(1);
      if (more) issue_part(lt, lpar, lchunk, std::integral_constant<int, 2>{});
      if (V3D_C0_ABLATE != 1) mfma_ky(2);
      PHASE_MARK(5);
    }

    // epilogue: rows 0-7 = x shift 0, rows 8-15 = x shift 1: lane (kq, jn) holds channels 4 (kq & 1) .. +3 of voxel
    // x = 2 jn + (kq >> 1).
    const int n = cur.n, oz0 = cur.oz0, oy0 = cur.oy0, ox0 = cur.ox0;
    if constexpr (SPLIT_OUT) {
      // split layout: those 4 channels are one 8-byte half of the voxel's hi slot and of its lo slot; the 4 lane
      // quarters of a wave instruction cover 16 x 2 consecutive slots completely -> direct stores, no LDS round trip
      const size_t out_plane_s = (size_t)p.Do * p.Ho * p.Wo;
      const int sx = kq >> 1;
      float bias[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[r] = lane_select(kq & 1, sbias[4 + r], sbias[r]);
      int half = kq & 1;
      asm volatile("" : "+v"(half));      // keeps this per-lane 64-bit base out of the tile loop's live registers (it was spilled)
      u32x2* const outs = reinterpret_cast<u32x2*>(p.out) + ((size_t)n * 2 * out_plane_s) * 2 + half;
      const int gx = ox0 + 2 * jn + sx;
      if (jn < C0::TW / 2 && gx < p.Wo) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
          const int gz = oz0 + j / RPW, gy = oy0 + wave * RPW + j % RPW;
          if (gz >= p.Do || gy >= p.Ho) continue;
          unsigned h[4], l[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float val = acc[j][r] + bias[r];
            if (p.relu) val = fmaxf(val, 0.f);
            h[r] = bf16_rne(val);
            l[r] = bf16_rne(val - __uint_as_float(h[r] << 16));
          }
          size_t sp = ((size_t)gz * p.Ho + gy) * p.Wo + gx;
          if (V3D_C0_ABLATE == 4 && p.relu != 12345) continue;        // everything computed, nothing stored (runtime-false guard)
          if (V3D_C0_ABLATE == 5) sp &= 0xffff;                       // every store lands in a 2 MB window (L2-resident)
          outs[sp * 2] = (u32x2){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
          outs[(out_plane_s + sp) * 2] = (u32x2){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
        }
      }
      PHASE_MARK(6);
    } else {
      // fp32 output: the tile goes through LDS ([co][z][y][28 x], co stride padded by 4 floats against bank conflicts)
      // so that it leaves as 16-byte row segments instead of 32 4-byte stores per lane.
      constexpr int OCS = C0::TD * C0::TH * C0::TW + 4;
      float* const os = reinterpret_cast<float*>(smem);
      __syncthreads();                 // every wave is done reading the input tile
      {
        const int sx = kq >> 1, cbase = 4 * (kq & 1);
        float bias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = lane_select(kq & 1, sbias[4 + r], sbias[r]);
        if (jn < C0::TW / 2) {
#pragma unroll
          for (int j = 0; j < NACC; ++j) {
            const int row = (j / RPW) * C0::TH + wave * RPW + j % RPW;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float val = acc[j][r] + bias[r];
              if (p.relu) val = fmaxf(val, 0.f);
              os[(cbase + r) * OCS + row * C0::TW + 2 * jn + sx] = val;
            }
          }
        }
      }
      __syncthreads();
      const size_t out_plane = (size_t)p.Do * p.Ho * p.Wo;
      constexpr int QPR = C0::TW / 4, NQ = 8 * C0::TD * C0::TH * QPR;      // float4 per row, per tile
#pragma unroll
      for (int k = 0; k < (NQ + NT - 1) / NT; ++k) {
        const int i = k * NT + tid;
        if (i >= NQ) break;
        const int co = i / (C0::TD * C0::TH * QPR), rem = i % (C0::TD * C0::TH * QPR);
        const int row = rem / QPR, q = rem % QPR;
        const int gz = oz0 + row / C0::TH, gy = oy0 + row % C0::TH, gx = ox0 + 4 * q;
        if (gz >= p.Do || gy >= p.Ho || gx >= p.Wo) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(os + co * OCS + row * C0::TW + 4 * q);
        const size_t o = ((size_t)n * 8 + co) * out_plane + ((size_t)gz * p.Ho + gy) * p.Wo + gx;
        if (gx + 3 < p.Wo && (p.Wo & 3) == 0) {
          if (p.skip) { const f32x4 sk = *reinterpret_cast<const f32x4*>(p.skip + o); v += sk; }
          *reinterpret_cast<f32x4*>(p.out + o) = v;
        } else {
          for (int e = 0; e < 4 && gx + e < p.Wo; ++e) p.out[o + e] = v[e] + (p.skip ? p.skip[o + e] : 0.f);
        }
      }
      PHASE_MARK(6);
    }
    cur = nxt;
  }
  PHASE_FLUSH;
}

// ---- conv1 .. conv6 on split-bf16 matrix cores -------------------------------------------------------------------
// The stride-1 / stride-2 layers behind conv0 read their input in the split channel-last layout ([n][CIN/8 groups]
// [hi, lo][D][H][W] 16-byte slots of 8 channels), so staging is plain 16-byte copies.  One workgroup = one output tile
// x 16 output channels (blockIdx.y picks the channel group).  The input channels are consumed 8 at a time: a chunk's
// halo'd tile and its 9 (kz, ky) weight fragments sit in LDS while the next chunk is already in registers.  MFMA
// tile: 16 rows = output channels, 16 columns = one output x row of 14 (or two rows of 8 on the coarsest level),
// K = 32 = 4 x taps x 8 channels (tap 3 carries zero weights).  Wave w owns output row(s) y of every plane of the
// tile; an input row read from LDS feeds every plane it contributes to.  Output: fp32 [n, COUT, D, H, W] (for the
// per-layer kernels that still follow), the split layout (for the next layer of this kind), or both.
enum { kOutF32 = 1, kOutSplit = 2 };

// FLAT (round 4, PropagationNet): no taps along z -- the "volume" is a stack of independent images [n][C/8][hi, lo][B][H][W]
// and the layer a batched 3x3 conv2d (upsampling.py:6-11); the weight image then holds the 3 ky fragments only.
template <int CIN_, int COUT_, int STRIDE_, int WB_, int OUT_, bool FLAT_ = false>
struct CG {
  static constexpr int CIN = CIN_, COUT = COUT_, S = STRIDE_, WB = WB_, OUT = OUT_;
  static constexpr bool FLAT = FLAT_;
  static constexpr int NCH = CIN / 8, NCG = COUT / 16;
  static constexpr int NRB = 16 / WB;                                  // output rows per MFMA column block
#ifndef V3D_FLAT_TD
#define V3D_FLAT_TD 8        // images per tile of the FLAT (conv2d) layers (developer A/B)
#endif
#ifndef V3D_CG_TD_S1
#define V3D_CG_TD_S1 4       // output planes per tile, stride-1 layers at 14-wide rows (developer A/B)
#endif
#ifndef V3D_CG_TD_S2
#define V3D_CG_TD_S2 3       // ... stride-2 layers (2 -> 3: conv1 0.207 -> 0.186 ms, conv3 0.116 -> 0.103; 4 leaves one workgroup per CU: 0.255)
#endif
  static constexpr int TD = FLAT ? V3D_FLAT_TD : WB != 14 ? (S == 2 ? 2 : 4) : S == 2 ? V3D_CG_TD_S2 : V3D_CG_TD_S1, TH = 4 * NRB, TW = WB;
  static constexpr int NKZ = FLAT ? 1 : 3;                             // z taps
  static constexpr int ID = FLAT ? TD : S * (TD - 1) + 3, IH = S * (TH - 1) + 3, IW = S * (TW - 1) + 3;
  static constexpr int NVOX = ID * IH * IW, NVOXP = NVOX + 8;         // idle lanes read a few slots past a row
  static constexpr int WQ = NKZ * 3 * 2 * 64;                          // 16-byte words of one chunk's weight image
  static constexpr int NVO = TD * TH * TW;
  static constexpr int LPR = IW <= 16 ? 16 : 32, RPI = 256 / LPR;      // staging: lanes per row, rows per iteration
  static constexpr int NROWS = 2 * ID * IH, NIT = (NROWS + RPI - 1) / RPI;
  static constexpr int NWIT = (WQ + 255) / 256;
  static constexpr size_t LDS_BYTES = (size_t)(2 * NVOXP + WQ) * 16;
  // Only the single-chunk layer (conv1) walks several items per workgroup: with one tile per workgroup it had no prefetch
  // at all (0.26 -> 0.22 ms).  On the multi-chunk layers the staging registers of the next item, live across the epilogue,
  // cost a workgroup per CU (conv2: 138 -> 195 VGPRs, 0.20 -> 0.25 ms), so they keep one item per workgroup.
  static constexpr bool PERSIST = NCH == 1;
  static constexpr int OCC = 2;      // (cutting conv2 / conv6 to 128 VGPRs for a fourth workgroup per CU: no gain / slower)
  static_assert(CIN % 8 == 0 && COUT % 16 == 0 && (WB == 14 || WB == 8) && (!FLAT || S == 1), "shape");
  // (the output staging starts at the LDS base: behind the last barrier the weight fragments are as dead as the input tile)
  static_assert((size_t)NVO * 64 <= LDS_BYTES, "split output staging fits in the workgroup's LDS");
  static_assert((size_t)16 * (NVO + 2) * 4 <= LDS_BYTES, "fp32 output staging fits in the workgroup's LDS");
  static_assert((NROWS + RPI) * ID * IH < (1 << 20) && 4 * NVO * NVO < (1 << 20), "v3d::small_div ranges of the staging / output index");
};

struct ConvGParams {
  const void* in;      // split layout, CIN / 8 groups
  const void* wp;      // [NCG][NCH][9][hi, lo][64 lanes][4 words]
  const float* bias;   // [COUT]
  float* out_f32;      // [n, COUT, Do, Ho, Wo] or null
  void* out_split;     // split layout, COUT / 8 groups, or null
  int n, Di, Hi, Wi, Do, Ho, Wo, ntz, nty, ntx;
  unsigned m_tx, m_ty, m_tz;   // v3d::magic_u32() of ntx, nty, ntz for the item index (0: divide)
};

// PERSISTENT: the grid is the number of workgroups the chip holds; a workgroup walks (tile, output channel group) items
// first + i, first + i + G, ... of its XCD's contiguous run (v3d::xcd_tile_walk; the channel group is the fastest index, so
// the workgroups that read the same input tile run side by side).  These layers have 1 to 8 chunks per tile: with one tile
// per workgroup, conv1 (one chunk) had no prefetch at all -- load, wait, compute, store, exit.  Where C::PERSIST is set the
// first chunk of the next item is requested before the MFMAs of the current item's last chunk; elsewhere the walk has one
// item per workgroup.
// `bias_r` = p.bias as a __restrict__ kernel argument: read inside the item loop after the previous item's stores, it can
// only stay a scalar load if the compiler can exclude that the outputs alias it.
template <class C>
__global__ __launch_bounds__(256, C::OCC) void convg_bf16x2_kernel(ConvGParams p, const float* __restrict__ bias_r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* const xs = reinterpret_cast<u32x4*>(smem);                 // [hi, lo][NVOXP] slots of the current chunk
  u32x4* const wq = xs + 2 * C::NVOXP;                              // [9][hi, lo][64] weight fragments
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  const size_t in_plane = (size_t)p.Di * p.Hi * p.Wi;
  const size_t out_plane = (size_t)p.Do * p.Ho * p.Wo;

  struct Item { int cg, n, oz0, oy0, ox0; };
  auto decode = [&](int t) __attribute__((always_inline)) {      // (wave-uniform: host magic numbers keep it on the scalar unit)
    Item q;
    const unsigned t0 = (unsigned)t / C::NCG;
    q.cg = (int)((unsigned)t - t0 * C::NCG);
    const unsigned t1 = v3d::udiv_magic(t0, (unsigned)p.ntx, p.m_tx), t2 = v3d::udiv_magic(t1, (unsigned)p.nty, p.m_ty);
    const unsigned t3 = v3d::udiv_magic(t2, (unsigned)p.ntz, p.m_tz);
    q.ox0 = (int)(t0 - t1 * (unsigned)p.ntx) * C::TW;
    q.oy0 = (int)(t1 - t2 * (unsigned)p.nty) * C::TH;
    q.oz0 = (int)(t2 - t3 * (unsigned)p.ntz) * C::TD;
    q.n = (int)t3;
    return q;
  };
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n * p.ntz * p.nty * p.ntx * C::NCG);
  if (walk.t >= walk.end) return;

  // ---- staging: chunk c = input channel group c (hi rows then lo rows) + its weight image -----------------------
  const int lrow = tid / C::LPR, lx = tid % C::LPR;
  const bool xok = lx < C::IW;
  u32x4 pre[C::NIT], wreg[C::NWIT];
  constexpr bool XPRE = C::PERSIST;
  auto issue = [&](const Item& q, int chunk) __attribute__((always_inline)) {
    const int iz0 = C::FLAT ? q.oz0 : C::S * q.oz0 - 1, iy0 = C::S * q.oy0 - 1, sgx = C::S * q.ox0 - 1 + lx;
    const bool xin = xok && sgx >= 0 && sgx < p.Wi;
    const int sxc = min(max(sgx, 0), p.Wi - 1);
    // (wave-uniform 64-bit base of the chunk's hi plane + a 32-bit lane offset built from 24-bit multiplies: with size_t
    // indices every row cost a v_mad_u64_u32 and two v_mul_lo_u32, quarter-rate instructions; the host checks the range)
    const char* const ins = reinterpret_cast<const char*>(reinterpret_cast<const u32x4*>(p.in) +
                                                          ((size_t)q.n * C::NCH + chunk) * 2 * in_plane);
    const u32x4* const wg = reinterpret_cast<const u32x4*>(p.wp) + (size_t)q.cg * C::NCH * C::WQ;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const unsigned rr = (unsigned)(it * C::RPI + lrow);
      const unsigned part = v3d::small_div<C::ID * C::IH>(rr), rem = rr - part * (C::ID * C::IH);
      const unsigned rz = v3d::small_div<C::IH>(rem), ry = rem - rz * C::IH;
      const int gz = iz0 + (int)rz, gy = iy0 + (int)ry;
      const bool ok = rr < C::NROWS && xin && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
      const int zc = min(max(gz, 0), p.Di - 1), yc = min(max(gy, 0), p.Hi - 1);
      const unsigned idx = (part ? (unsigned)in_plane : 0u) + __umul24(__umul24(zc, p.Hi) + yc, p.Wi) + sxc;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ins + idx * 16u);
      pre[it] = ok ? v : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < C::NWIT; ++i)
      wreg[i] = (i * 256 + tid < C::WQ) ? wg[(size_t)chunk * C::WQ + i * 256 + tid] : (u32x4){0u, 0u, 0u, 0u};
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const unsigned rr = (unsigned)(it * C::RPI + lrow);
      const unsigned part = v3d::small_div<C::ID * C::IH>(rr), rem = rr - part * (C::ID * C::IH);
      if (rr < C::NROWS && xok) xs[part * C::NVOXP + rem * C::IW + lx] = pre[it];
    }
#pragma unroll
    for (int i = 0; i < C::NWIT; ++i)
      if (i * 256 + tid < C::WQ) wq[i * 256 + tid] = wreg[i];
  };

  // ---- MFMA role: wave w = output rows y = w * NRB + r, lane column jn = (r, x) ----------------------------------
  const int lr = jn / C::WB, lxo = jn % C::WB;
  f32x4 acc[C::TD];
  const u32x4* const wf = wq + lane;
  const int ly = wave * C::NRB + lr;                                   // this lane's output row inside the tile
  const bool live = jn < C::NRB * C::WB;                               // lanes 14, 15 of a 14-wide column block idle

  PHASE_DECL;
  Item cur = decode(walk.t);
  issue(cur, 0);
  PHASE_MARK(0);
#pragma unroll 1
  for (int t = walk.t; t < walk.end; t += walk.step) {
    const int tn = t + walk.step;
    const bool has_next = C::PERSIST && tn < walk.end;
    const Item nxt = decode(has_next ? tn : t);
#pragma unroll
    for (int z = 0; z < C::TD; ++z) acc[z] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int chunk = 0; chunk < C::NCH; ++chunk) {
      __syncthreads();                 // the previous chunk's MFMAs / the previous item's output staging are done with the LDS
      commit();
      if (chunk == 0 && tid < 16) xs[(tid >> 3) * C::NVOXP + C::NVOX + (tid & 7)] = (u32x4){0u, 0u, 0u, 0u};   // pad slots stay finite
      __syncthreads();
      PHASE_MARK(1);
      if (chunk + 1 < C::NCH) issue(cur, chunk + 1);
      else if (XPRE && has_next) issue(nxt, 0);
#ifndef V3D_CG_KYU
#define V3D_CG_KYU 1
#endif
#pragma unroll V3D_CG_KYU
      for (int ky = 0; ky < 3; ++ky) {
        bf16x8 a_hi[C::NKZ], a_lo[C::NKZ];
#pragma unroll
        for (int kz = 0; kz < C::NKZ; ++kz) {
          a_hi[kz] = __builtin_bit_cast(bf16x8, wf[((kz * 3 + ky) * 2) * 64]);
          a_lo[kz] = __builtin_bit_cast(bf16x8, wf[((kz * 3 + ky) * 2 + 1) * 64]);
        }
        const int rowbase = (C::S * (wave * C::NRB + lr) + ky) * C::IW + C::S * lxo + kq;
#pragma unroll
        for (int iz = 0; iz < C::ID; ++iz) {
          const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xs[rowbase + iz * C::IH * C::IW]);
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xs[rowbase + iz * C::IH * C::IW + C::NVOXP]);
#pragma unroll
          for (int kz = 0; kz < C::NKZ; ++kz) {
            const int z2 = iz - kz;      // (FLAT: the single tap is the centre one: input plane iz -> output plane iz)
            if (z2 >= 0 && z2 % C::S == 0 && z2 / C::S < C::TD)
              acc[z2 / C::S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[kz], b_hi, acc[z2 / C::S], 0, 0, 0);
          }
#pragma unroll
          for (int kz = 0; kz < C::NKZ; ++kz) {
            const int z2 = iz - kz;
            if (z2 >= 0 && z2 % C::S == 0 && z2 / C::S < C::TD)
              acc[z2 / C::S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[kz], b_lo, acc[z2 / C::S], 0, 0, 0);
          }
#pragma unroll
          for (int kz = 0; kz < C::NKZ; ++kz) {
            const int z2 = iz - kz;
            if (z2 >= 0 && z2 % C::S == 0 && z2 / C::S < C::TD)
              acc[z2 / C::S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[kz], b_hi, acc[z2 / C::S], 0, 0, 0);
          }
        }
      }
      PHASE_MARK(2);
    }
    __syncthreads();                 // the input tile is dead: its LDS stages the output tile
    PHASE_MARK(3);

    // ---- bias + ReLU, through LDS, out in 16-byte (split layout) and / or 8-byte (fp32 rows) pieces ---------------
    const int cg = cur.cg, n = cur.n, oz0 = cur.oz0, oy0 = cur.oy0, ox0 = cur.ox0;
    float vout[C::TD][4];
    {
      // the bias comes through SGPRs (a per-lane vector load here would make the epilogue wait for the prefetched tile)
      float bias[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b0 = bias_r[cg * 16 + r], b1 = bias_r[cg * 16 + 4 + r], b2 = bias_r[cg * 16 + 8 + r], b3 = bias_r[cg * 16 + 12 + r];
        bias[r] = lane_select(kq & 2, lane_select(kq & 1, b3, b2), lane_select(kq & 1, b1, b0));
      }
#pragma unroll
      for (int z = 0; z < C::TD; ++z)
#pragma unroll
        for (int r = 0; r < 4; ++r) vout[z][r] = fmaxf(acc[z][r] + bias[r], 0.f);
    }
    if constexpr ((C::OUT & kOutSplit) != 0) {
      u32x2* const sp2 = reinterpret_cast<u32x2*>(smem);               // [2 groups][hi, lo][NVO] slots, 8-byte halves
#pragma unroll
      for (int z = 0; z < C::TD; ++z) {
        if (!live) break;
        u32x2 hp, lp;
        split4(vout[z][0], vout[z][1], vout[z][2], vout[z][3], hp, lp);
        const int vox = (z * C::TH + ly) * C::TW + lxo;
        sp2[((((kq >> 1) * 2 + 0) * C::NVO) + vox) * 2 + (kq & 1)] = hp;
        sp2[((((kq >> 1) * 2 + 1) * C::NVO) + vox) * 2 + (kq & 1)] = lp;
      }
      __syncthreads();
      char* const outs = reinterpret_cast<char*>(reinterpret_cast<u32x4*>(p.out_split) + ((size_t)n * (C::COUT / 8) + cg * 2) * 2 * out_plane);
      const u32x4* const sq = reinterpret_cast<const u32x4*>(smem);
      for (int i = tid; i < 4 * C::NVO; i += 256) {
        const unsigned gp = v3d::small_div<C::NVO>((unsigned)i), vox = (unsigned)i - gp * C::NVO;
        const unsigned vz = v3d::small_div<C::TH * C::TW>(vox), vr = vox - vz * (C::TH * C::TW);
        const unsigned vy = v3d::small_div<C::TW>(vr), vx = vr - vy * C::TW;
        const int gz = oz0 + (int)vz, gy = oy0 + (int)vy, gx = ox0 + (int)vx;
        // (32-bit slot index on the wave-uniform base of the item's first plane, as in the staging loads)
        const unsigned idx = __umul24(gp, (unsigned)out_plane) + __umul24(__umul24(gz, p.Ho) + gy, p.Wo) + gx;
        if (gz < p.Do && gy < p.Ho && gx < p.Wo) *reinterpret_cast<u32x4*>(outs + idx * 16u) = sq[i];
      }
      if constexpr ((C::OUT & kOutF32) != 0) __syncthreads();
    }
    if constexpr ((C::OUT & kOutF32) != 0) {
      constexpr int OCS = C::NVO + 2;
      float* const os = reinterpret_cast<float*>(smem);                // [16 co][NVO (+2)]
      if (live) {
#pragma unroll
        for (int z = 0; z < C::TD; ++z)
#pragma unroll
          for (int r = 0; r < 4; ++r) os[(4 * kq + r) * OCS + (z * C::TH + ly) * C::TW + lxo] = vout[z][r];
      }
      __syncthreads();
      constexpr int NR = C::TD * C::TH, QPR = C::TW / 2;
      for (int i = tid; i < 16 * NR * QPR; i += 256) {
        const int co = i / (NR * QPR), row = (i / QPR) % NR, q = i % QPR;
        const int gz = oz0 + row / C::TH, gy = oy0 + row % C::TH, gx = ox0 + 2 * q;
        if (gz >= p.Do || gy >= p.Ho || gx >= p.Wo) continue;
        const float2 v = *reinterpret_cast<const float2*>(os + co * OCS + row * C::TW + 2 * q);
        float* o = p.out_f32 + ((size_t)n * C::COUT + cg * 16 + co) * out_plane + ((size_t)gz * p.Ho + gy) * p.Wo + gx;
        if (gx + 1 < p.Wo && (p.Wo & 1) == 0) *reinterpret_cast<float2*>(o) = v;
        else { o[0] = v.x; if (gx + 1 < p.Wo) o[1] = v.y; }
      }
    }
    PHASE_MARK(4);
    if (!C::PERSIST) break;
    cur = nxt;
  }
  PHASE_FLUSH;
}

// ---- conv7 / conv8 (transposed convolutions + skip) on split-bf16 matrix cores -------------------------------------
// ConvTranspose3d(k 3, stride 2, pad 1, output_padding 1): per axis out[2j] = in[j] w[1], out[2j+1] = in[j] w[2] +
// in[j+1] w[0].  Cell j = outputs (2j, 2j+1) from inputs (j, j+1).  GEMM per chunk of 8 input channels: columns = the
// cells of an x row, rows = 16 output channels of one of the 8 output parities, K = 32 = (dy, dx) x 8 channels; the
// (pz, dz) combinations (0,0), (1,0), (1,1) x 4 (py, px) parities = 12 blocks of weights (zero where a parity does not
// see an input).  A wave keeps the B fragments of its two cell rows in registers and walks the 12 blocks.
// Input: split channel-last layout; output = ReLU(BN(deconv)) + skip (fp32 [n, COUT, D, H, W]) as fp32 or split.
// SKIP_SPLIT (the fused chain): the skip tensor comes in the split layout its producer (conv2 / conv4) hands to the next encoder
// layer anyway -- skip = hi + lo, the 16 mantissa bits every other operand of this mode has, as conv9+prob reads the conv0
// skip -- so that the encoder layer does not write a second, fp32 copy of its output (conv2: 154 of 463 MB per 64 cfg2 views).
template <int CIN_, int COUT_, int WB_, int OUT_, bool SKIP_SPLIT_ = false>
struct DG {
  static constexpr int CIN = CIN_, COUT = COUT_, WB = WB_, OUT = OUT_;
  static constexpr bool SKIP_SPLIT = SKIP_SPLIT_;
  static constexpr int NCH = CIN / 8, NCG = COUT / 16;
  static constexpr int NRB = 16 / WB;                         // cell rows per MFMA column block
  static constexpr int CZ = 2, CY = 4 * NRB, CX = WB;         // cells per tile; wave w = cell rows w*NRB .. of both cz
  static constexpr int VZ = CZ + 1, VY = CY + 1, VX = CX + 2; // input voxels (+1 idle column)
  static constexpr int NVOX = VZ * VY * VX, NVOXP = NVOX + 8;
  static constexpr int WQ = 12 * 2 * 64;                      // 16-byte words of one chunk's weight image
  static constexpr int NSLOT = 2 * NVOX;
  static constexpr int NIT = (NSLOT + 255) / 256, NWIT = (WQ + 255) / 256;
  static constexpr int LDS_BYTES = (2 * NVOXP + WQ) * 16;
  static_assert(CIN % 8 == 0 && COUT % 16 == 0 && (WB == 14 || WB == 8), "shape");
  static_assert(LDS_BYTES <= 64 * 1024, "static LDS");
};

struct DeconvGParams {
  const void* in;      // split layout [n][CIN/8][hi, lo][Di][Hi][Wi]
  const void* wp;      // [NCG][NCH][12][hi, lo][64 lanes][4 words]
  const float* bias;   // [COUT]
  const float* skip;   // [n, COUT, 2Di, 2Hi, 2Wi] fp32 -- or, SKIP_SPLIT, the split layout [n][COUT/8][hi, lo][2Di][2Hi][2Wi]
  float* out_f32;      // same shape, or null
  void* out_split;     // split layout, or null
  int n, Di, Hi, Wi, ntz, nty, ntx;
  unsigned m_tx, m_ty, m_tz;   // v3d::magic_u32() of ntx, nty, ntz for the tile index (0: divide)
};

template <class C>
__global__ __launch_bounds__(256, 2) void deconvg_bf16x2_kernel(DeconvGParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
  u32x4* const xs = reinterpret_cast<u32x4*>(smem);                 // [hi, lo][NVOXP]
  u32x4* const wq = xs + 2 * C::NVOXP;                              // [12][hi, lo][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  const int cg = blockIdx.y;
  const unsigned b0 = (unsigned)v3d::xcd_contiguous_block();
  const unsigned b1 = v3d::udiv_magic(b0, (unsigned)p.ntx, p.m_tx), b2 = v3d::udiv_magic(b1, (unsigned)p.nty, p.m_ty);
  const unsigned b3 = v3d::udiv_magic(b2, (unsigned)p.ntz, p.m_tz);
  const int tx = (int)(b0 - b1 * (unsigned)p.ntx), ty = (int)(b1 - b2 * (unsigned)p.nty), tz = (int)(b2 - b3 * (unsigned)p.ntz);
  const int n = (int)b3;
  const int cz0 = tz * C::CZ, cy0 = ty * C::CY, cx0 = tx * C::CX;   // first cell = first input voxel of the tile
  const int Do = 2 * p.Di, Ho = 2 * p.Hi, Wo = 2 * p.Wi;
  const size_t in_plane = (size_t)p.Di * p.Hi * p.Wi;
  const size_t out_plane = (size_t)Do * Ho * Wo;

  const u32x4* const ins = reinterpret_cast<const u32x4*>(p.in) + (size_t)n * C::NCH * 2 * in_plane;
  const u32x4* const wg = reinterpret_cast<const u32x4*>(p.wp) + (size_t)cg * C::NCH * C::WQ;
  u32x4 pre[C::NIT], wreg[C::NWIT];
  static_assert(C::NIT * 256 * C::NVOX < (1 << 20), "v3d::small_div range of the staging index");
  auto issue = [&](int chunk) __attribute__((always_inline)) {
    // (32-bit slot offsets on the chunk's wave-uniform base, as in convg_bf16x2_kernel; the host checks the range)
    const char* const insc = reinterpret_cast<const char*>(ins + (size_t)chunk * 2 * in_plane);
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const unsigned i = (unsigned)(it * 256 + tid);
      const unsigned part = v3d::small_div<C::NVOX>(i), v = i - part * C::NVOX;
      const unsigned vz = v3d::small_div<C::VY * C::VX>(v), vr = v - vz * (C::VY * C::VX);
      const unsigned vy = v3d::small_div<C::VX>(vr), vx = vr - vy * C::VX;
      const int gz = cz0 + (int)vz, gy = cy0 + (int)vy, gx = cx0 + (int)vx;
      const bool ok = i < C::NSLOT && gz < p.Di && gy < p.Hi && gx < p.Wi;       // inputs past the volume do not exist
      const int zc = min(gz, p.Di - 1), yc = min(gy, p.Hi - 1), xc = min(gx, p.Wi - 1);
      const unsigned idx = (part ? (unsigned)in_plane : 0u) + __umul24(__umul24(zc, p.Hi) + yc, p.Wi) + xc;
      const u32x4 val = *reinterpret_cast<const u32x4*>(insc + idx * 16u);
      pre[it] = ok ? val : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < C::NWIT; ++i)
      wreg[i] = (i * 256 + tid < C::WQ) ? wg[(size_t)chunk * C::WQ + i * 256 + tid] : (u32x4){0u, 0u, 0u, 0u};
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const unsigned i = (unsigned)(it * 256 + tid), part = v3d::small_div<C::NVOX>(i);
      if (i < C::NSLOT) xs[part * C::NVOXP + (i - part * C::NVOX)] = pre[it];
    }
#pragma unroll
    for (int i = 0; i < C::NWIT; ++i)
      if (i * 256 + tid < C::WQ) wq[i * 256 + tid] = wreg[i];
  };
  if (tid < 16) xs[(tid >> 3) * C::NVOXP + C::NVOX + (tid & 7)] = (u32x4){0u, 0u, 0u, 0u};

  // column jn = (cell row r within the wave's pair, cell x); K lane group kq = (dy, dx)
  const int lr = jn / C::WB, lxo = jn % C::WB;
  const int lcy = wave * C::NRB + lr;
  f32x4 acc[C::CZ][8];
#pragma unroll
  for (int z = 0; z < C::CZ; ++z)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[z][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const u32x4* const wf = wq + lane;

  issue(0);
#pragma unroll 1
  for (int chunk = 0; chunk < C::NCH; ++chunk) {
    __syncthreads();
    commit();
    __syncthreads();
    if (chunk + 1 < C::NCH) issue(chunk + 1);
    // B fragments of this wave's cell rows: input planes cz .. cz + 1 for cz = 0 .. CZ-1 => VZ planes
    bf16x8 b_hi[C::VZ], b_lo[C::VZ];
#pragma unroll
    for (int vz = 0; vz < C::VZ; ++vz) {
      const int slot = (vz * C::VY + lcy + (kq >> 1)) * C::VX + lxo + (kq & 1);
      b_hi[vz] = __builtin_bit_cast(bf16x8, xs[slot]);
      b_lo[vz] = __builtin_bit_cast(bf16x8, xs[slot + C::NVOXP]);
    }
#pragma unroll
    for (int pyx = 0; pyx < 4; ++pyx) {
#pragma unroll
      for (int zb = 0; zb < 3; ++zb) {                        // (pz, dz) = (0,0), (1,0), (1,1)
        const int pz = zb == 0 ? 0 : 1, dz = zb == 2 ? 1 : 0;
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, wf[((zb * 4 + pyx) * 2) * 64]);
        const bf16x8 a_lo = __builtin_bit_cast(bf16x8, wf[((zb * 4 + pyx) * 2 + 1) * 64]);
#pragma unroll
        for (int cz = 0; cz < C::CZ; ++cz) {
          f32x4& c = acc[cz][pz * 4 + pyx];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi[cz + dz], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo[cz + dz], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi[cz + dz], c, 0, 0, 0);
        }
      }
    }
  }

  // ---- BN bias + ReLU + skip; the two x parities of a cell are one float2 / adjacent slots ----------------------------
  const bool live = jn < C::NRB * C::WB;
  const int gcy = cy0 + lcy, gcx = cx0 + lxo;
  if (live && gcy < p.Hi && gcx < p.Wi) {
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = p.bias[cg * 16 + 4 * kq + r];
#pragma unroll
    for (int cz = 0; cz < C::CZ; ++cz) {
      const int gcz = cz0 + cz;
      if (gcz >= p.Di) break;
#pragma unroll
      for (int pzy = 0; pzy < 4; ++pzy) {
        const int pz = pzy >> 1, py = pzy & 1;
        const size_t sp = ((size_t)(2 * gcz + pz) * Ho + (2 * gcy + py)) * Wo + 2 * gcx;
        float v0[4], v1[4];
        float sk0[4], sk1[4];
        if constexpr (C::SKIP_SPLIT) {
          // channels 4 kq .. 4 kq + 3 of group cg * 2 + (kq >> 1): one 8-byte half of the hi / lo slots of the two voxels
          const u32x2* const ss = reinterpret_cast<const u32x2*>(p.skip) +
                                  (((size_t)n * (C::COUT / 8) + cg * 2 + (kq >> 1)) * 2 * out_plane) * 2 + (kq & 1);
          const u32x2 h0 = ss[sp * 2], h1 = ss[(sp + 1) * 2], l0 = ss[(out_plane + sp) * 2], l1 = ss[(out_plane + sp + 1) * 2];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned sh = (r & 1) ? 0u : 16u, mk = (r & 1) ? 0xffff0000u : 0xffffffffu;
            sk0[r] = __uint_as_float((h0[r >> 1] << sh) & mk) + __uint_as_float((l0[r >> 1] << sh) & mk);
            sk1[r] = __uint_as_float((h1[r >> 1] << sh) & mk) + __uint_as_float((l1[r >> 1] << sh) & mk);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t o = ((size_t)n * C::COUT + cg * 16 + 4 * kq + r) * out_plane + sp;
          if constexpr (!C::SKIP_SPLIT) {
            const float2 sk = *reinterpret_cast<const float2*>(p.skip + o);
            sk0[r] = sk.x; sk1[r] = sk.y;
          }
          v0[r] = fmaxf(acc[cz][pz * 4 + py * 2 + 0][r] + bias[r], 0.f) + sk0[r];
          v1[r] = fmaxf(acc[cz][pz * 4 + py * 2 + 1][r] + bias[r], 0.f) + sk1[r];
          if constexpr ((C::OUT & kOutF32) != 0) *reinterpret_cast<float2*>(p.out_f32 + o) = make_float2(v0[r], v1[r]);
        }
        if constexpr ((C::OUT & kOutSplit) != 0) {
          // this lane holds channels 4 kq .. 4 kq + 3 of group cg * 2 + (kq >> 1): one 8-byte half of the hi / lo slot
          u32x2* const os = reinterpret_cast<u32x2*>(p.out_split) +
                            (((size_t)n * (C::COUT / 8) + cg * 2 + (kq >> 1)) * 2 * out_plane) * 2 + (kq & 1);
          u32x2 h0, l0, h1, l1;
          split4(v0[0], v0[1], v0[2], v0[3], h0, l0);
          split4(v1[0], v1[1], v1[2], v1[3], h1, l1);
          os[sp * 2] = h0;
          os[(sp + 1) * 2] = h1;
          os[(out_plane + sp) * 2] = l0;
          os[(out_plane + sp + 1) * 2] = l1;
        }
      }
    }
  }
}

// ---- prob conv (base -> 1 channel, bias, no BN/ReLU; mvsnet.py:152,162) ---------------------------
// A 1-channel output would waste 15/16 of an MFMA, so this layer is register-blocked VALU work:
// a workgroup owns a PT_D x PT_H x (PT_XG*PT_RX) output tile; CK input channels of the halo'd tile
// are staged in LDS (row stride == 8 (mod 32) floats so the 8 x-groups x 4 rows of a half-wave hit 32
// distinct banks); each thread produces PT_RX consecutive x outputs from a sliding 9-float window,
// weights come in through the scalar cache (wave-uniform addresses).
constexpr int PT_D = 4, PT_H = 8, PT_XG = 8, PT_RX = 7, PT_W = PT_XG * PT_RX;   // 4 x 8 x 56 tile
constexpr int PT_CK = 2;
constexpr int PT_ID = PT_D + 2, PT_IH = PT_H + 2, PT_IW = PT_W + 2, PT_RS = 72;
constexpr int PT_PLANE = PT_ID * PT_IH * PT_RS;

// The halo'd rows are staged as aligned float4s [ox0 - 4, ox0 + 60) -- 16 lanes per row, 16 rows per workgroup instruction
// round (the regulariser's volumes are multiples of 8 wide, v3d_costreg_depth_f32) -- and the window of an output starts 3
// floats into the LDS row.  With one float per lane (58 of 64 lanes, 232 bytes per wave instruction at the 4-byte rate of the
// CU's vector-memory path) the staging loads were the kernel: 0.62 ms per 64 views at cfg2 against 0.1 ms of arithmetic.
template <int CIN>
__global__ __launch_bounds__(256) void prob_conv_kernel(const float* __restrict__ in,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out, int n, int D, int H,
                                                        int W, int ntz, int nty, int ntx) {
  __shared__ __attribute__((aligned(16))) float xs[PT_CK * PT_PLANE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = v3d::xcd_contiguous_block();     // neighbouring tiles (shared halo) on the same XCD's L2
  const int tx = b % ntx; b /= ntx;
  const int ty = b % nty; b /= nty;
  const int tz = b % ntz;
  const int bn = b / ntz;
  const int oz0 = tz * PT_D, oy0 = ty * PT_H, ox0 = tx * PT_W;
  const size_t plane = (size_t)D * H * W;
  const float* inb = in + (size_t)bn * CIN * plane;

  // compute role: one z per wave, lane = y * 8 + xg
  const int cz = wave, cy = lane >> 3, cxg = lane & 7;
  const int cbase = (cz * PT_IH + cy) * PT_RS + cxg * PT_RX + 3;
  float acc[PT_RX];
#pragma unroll
  for (int i = 0; i < PT_RX; ++i) acc[i] = 0.f;

  constexpr int ROWS = PT_CK * PT_ID * PT_IH;          // 16 lanes per row
  constexpr int NIT = (ROWS + 15) / 16;
  static_assert(PT_W % 4 == 0 && PT_RS % 4 == 0 && PT_RS >= 64, "float4 staging");
  // staging: the next chunk's rows are loaded into registers right after the barrier and stay in flight while the current
  // chunk is being consumed
  const int sgx = ox0 - 4 + 4 * (tid & 15);
  const bool xin = sgx >= 0 && sgx + 3 < W;
  f32x4 pre[NIT];
  auto issue = [&](int c0) {
    int gx_o = sgx;                           // opaque copies keep the per-row address arithmetic from
    asm volatile("" : "+v"(gx_o));            // being hoisted out of the chunk loop into VGPRs
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * 16 + (tid >> 4);
      const int ck = row / (PT_ID * PT_IH), rz = (row / PT_IH) % PT_ID, ry = row % PT_IH;
      const int gz = oz0 - 1 + rz, gy = oy0 - 1 + ry;
      const bool ok = row < ROWS && xin && gz >= 0 && gz < D && gy >= 0 && gy < H;
      const float* src = inb + (size_t)(c0 + ck) * plane + ((size_t)gz * H + gy) * W + gx_o;
      pre[it] = ok ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto commit = [&]() {
    int lane_o = 4 * (tid & 15);
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * 16 + (tid >> 4);
      const int ck = row / (PT_ID * PT_IH), rz = (row / PT_IH) % PT_ID, ry = row % PT_IH;
      if (row < ROWS) *reinterpret_cast<f32x4*>(xs + ck * PT_PLANE + (rz * PT_IH + ry) * PT_RS + lane_o) = pre[it];
    }
  };
  issue(0);
#pragma unroll 1
  for (int c0 = 0; c0 < CIN; c0 += PT_CK) {
    __syncthreads();
    commit();
    __syncthreads();
    if (c0 + PT_CK < CIN) issue(c0 + PT_CK);
#pragma unroll
    for (int ck = 0; ck < PT_CK; ++ck) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* row = xs + ck * PT_PLANE + cbase + (kz * PT_IH + ky) * PT_RS;
          float win[PT_RX + 2];
#pragma unroll
          for (int i = 0; i < PT_RX + 2; ++i) win[i] = row[i];
          const float* wk = w + (c0 + ck) * 27 + (kz * 3 + ky) * 3;   // wave-uniform -> s_load
          const float w0 = wk[0], w1 = wk[1], w2 = wk[2];
#pragma unroll
          for (int i = 0; i < PT_RX; ++i) acc[i] += win[i] * w0 + win[i + 1] * w1 + win[i + 2] * w2;
        }
      }
    }
  }
  const int gz = oz0 + cz, gy = oy0 + cy;
  if (gz < D && gy < H) {
    const float bsv = bias[0];
    float* o = out + (size_t)bn * plane + ((size_t)gz * H + gy) * W;
#pragma unroll
    for (int i = 0; i < PT_RX; ++i) {
      const int gx = ox0 + cxg * PT_RX + i;
      if (gx < W) o[gx] = acc[i] + bsv;
    }
  }
}

// ---- conv9 + prob fused (mvsnet.py:161-162): x_reg = prob(conv0 + ReLU(BN(deconv9(u8)))) --------------------
// The 8-channel full-resolution tensor between the last deconvolution and the 1-channel prob conv (308 MB per
// 32 references, written once and read once) never leaves the CU: a workgroup owns a 4 x 8 x 28 tile of the
// prob output, computes the deconvolution on the tile + 1 halo (6 x 10 x 30) with split-bf16 MFMAs, adds the
// conv0 skip, parks the result in LDS and runs the 3x3x3x8 prob conv from there.
//
// Deconvolution as a GEMM over "cells".  ConvTranspose3d(k=3, stride 2, pad 1, output_padding 1) gives
// out[o] = sum_i in[i] w[o - 2i + 1], so per axis output 2j+1 = in[j] w[2] + in[j+1] w[0] and output 2j+2 =
// in[j+1] w[1].  Cell j therefore owns outputs (2j+1, 2j+2) and reads inputs (j, j+1); the halo'd tile starts
// at an odd output, i.e. it is exactly 3 x 5 x 15 cells.  One MFMA tile: 16 columns = the 15 cells of an x row
// (+1 idle), 16 rows = 2 x parities x 8 output channels, K = 32 = 2 x inputs x 16 input channels; the
// (z, y) parities / inputs are separate blocks, of which 9 of 16 hold weights (the 27 taps).
struct C9 {
  static constexpr int TD = 4, TH = 8, TW = 28;                    // prob output tile
  static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;      // u9 tile with halo
  static constexpr int CZ = HD / 2, CY = HH / 2, CX = HW / 2;      // 3 x 5 x 15 cells
  static constexpr int VZ = CZ + 1, VY = CY + 1, VX = CX + 2;      // 4 x 6 x (16 + 1 idle) input voxels
  static constexpr int NVOX = VZ * VY * VX;
  static constexpr int RS = 32;                                    // u9 tile row stride in floats
  static constexpr int U9_BYTES = 8 * HD * HH * RS * 4;            // 61440: [4 channel pairs][HD][HH][RS][2]
  static constexpr int IN_BYTES = NVOX * 2 * 16 * 2;               // hi + lo, two 8-channel halves per voxel
  static constexpr int LDS_BYTES = U9_BYTES > IN_BYTES ? U9_BYTES : IN_BYTES;
  static constexpr int NCB = CZ * CY;                              // 15 cell rows = MFMA column blocks
  static constexpr int WU32 = 9 * 2 * 64 * 4;                      // weight image words
  // exact-fp32 variant (conv9_prob_kernel<true>): the input tile as [16 channels][VZ][VY][VX] floats at a channel stride of
  // FS floats (== 16 mod 64: the four k lane groups of a B fragment read four channels' rows from four different bank
  // quarters), the weight fragments as [9 blocks][2 halves][64 lanes][4 k slices] floats
  static constexpr int FS = 464, WF32 = 9 * 64 * 8;
  static_assert(FS >= NVOX && FS % 64 == 16 && (16 * FS + WF32) * 4 <= U9_BYTES, "fp32 input tile + weights share the u9 region");
};

struct C9Params {
  const float* u8;     // conv8 output in the split layout [n][2 groups][hi, lo][D/2][H/2][W/2][8 bf16]   (F32: fp32 [n, 16, D/2, H/2, W/2])
  const float* c0;     // conv0 output (skip) in the split channel-last layout [n][hi, lo][D][H][W][8 bf16]  (F32: fp32 [n, 8, D, H, W])
  const float* wbf;    // split-bf16 fragment image of the deconv weights (BN scale folded)                 (F32: fp32 fragments, C9::WF32)
  const float* bias9;  // [8] folded BN bias
  const float* wprob;  // [4 channel pairs, 27, 2]
  const float* bprob;  // [1]
  float* out;          // [n, D, H, W]
  int n, D, H, W, ntz, nty, ntx;
  int zy_order;        // tile_order(): 1 = x, z, y
  unsigned m_tx, m_t1, m_t2;   // v3d::magic_u32() of ntx and of the two tile counts divided after it (in zy_order's order)
};

// Address arithmetic (round 4).  A workgroup lives for one tile, so its prologue -- tile index -> (n, tz, ty, tx), the clamped
// addresses of 3 input slots and 16 skip half-slots per lane -- is paid per tile: written with size_t indices it was 46
// quarter-rate integer instructions (v_mad_u64_u32, v_mul_lo_u32, five v_rcp_iflag_f32 division sequences) among ~850 VALU
// instructions of a wave, as many issue slots as the prob conv's FMAs.  Now: host magic numbers for the tile index, 24-bit
// multiplies, wave-uniform row bases in SGPRs + 32-bit lane offsets (the host checks that a view's tensors stay below 4 GB).
// 8 waves per workgroup, <= 128 VGPRs: two workgroups = 16 waves per CU (round 1's 4-wave version needed 218 VGPRs, i.e.
// 8 waves per CU, and spent most of its time waiting for its own loads, barriers and LDS round trips: 0.55 -> 0.45 ms).  The weight fragments
// are parked in LDS next to the input tile (both are dead before the u9 tile overwrites them) instead of 72 registers, a
// wave owns two of the 15 cell rows in the deconvolution, and in the prob conv a lane owns two consecutive outputs so that
// the 16 lanes of a row read 16 consecutive 16-byte slots (conflict-free; the 4-outputs-per-lane mapping read at a 32-byte
// lane stride: 2-way conflicts on every read).  Accumulation orders are those of the 4-wave kernel: bit-identical output.
#ifndef V3D_C9_ABLATE
#define V3D_C9_ABLATE 0      // developer ablations (scripts/ab_build.sh): 1 no prob loop, 2 one channel pair only, 3 no MFMAs, 4 no skip loads, 5 no input loads
#endif
// F32 (round 4, the exact-fp32 chain): the same tile structure on v_mfma_f32_16x16x4_f32 -- fp32 tensors in the reference
// layout in, eight k slices of 4 per block where the split path has three bf16 products of 32 -- instead of the generic
// per-layer conv9 kernel + prob_conv_kernel, which wrote and re-read the 8-channel full-resolution tensor (0.82 + 0.30 ms).
template <bool F32>
__global__ __launch_bounds__(512, 4) void conv9_prob_kernel(C9Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[C9::LDS_BYTES];
  u32x4* const xh = reinterpret_cast<u32x4*>(smem);             // [NVOX][2 halves] hi slots
  u32x4* const xl = xh + C9::NVOX * 2;                          // lo slots
  u32x4* const wfr = xl + C9::NVOX * 2;                         // [9 blocks][hi, lo][64 lanes] weight fragments
  float* const xf = reinterpret_cast<float*>(smem);             // F32: [16][FS] input tile
  float* const wff = xf + 16 * C9::FS;                          // F32: [9 blocks][2 halves][64 lanes][4 k slices] weight fragments
  float* const u9s = reinterpret_cast<float*>(smem);            // [8][HD][HH][RS], reuses the input tile
  static_assert(C9::IN_BYTES + C9::WU32 * 4 <= C9::U9_BYTES, "input tile + weight fragments share the u9 region");
  constexpr int NT = 512, NCBI = 2;                             // threads, cell rows per wave (15 rows over 8 waves)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  const unsigned b0 = (unsigned)v3d::xcd_contiguous_block();
  const unsigned b1 = v3d::udiv_magic(b0, (unsigned)p.ntx, p.m_tx);
  const int tx = (int)(b0 - b1 * (unsigned)p.ntx);
  int ty, tz, n;
  {
    const unsigned d1 = (unsigned)(p.zy_order ? p.ntz : p.nty), d2 = (unsigned)(p.zy_order ? p.nty : p.ntz);
    const unsigned b2 = v3d::udiv_magic(b1, d1, p.m_t1), b3 = v3d::udiv_magic(b2, d2, p.m_t2);
    const int t1 = (int)(b1 - b2 * d1), t2 = (int)(b2 - b3 * d2);
    tz = p.zy_order ? t1 : t2;
    ty = p.zy_order ? t2 : t1;
    n = (int)b3;
  }
  const int oz0 = tz * C9::TD, oy0 = ty * C9::TH, ox0 = tx * C9::TW;
  const int D2 = p.D >> 1, H2 = p.H >> 1, W2 = p.W >> 1;
  const int iz0 = (oz0 >> 1) - 1, iy0 = (oy0 >> 1) - 1, ix0 = (ox0 >> 1) - 1;
  const size_t in_plane = (size_t)D2 * H2 * W2;
  const size_t out_plane = (size_t)p.D * p.H * p.W;

  PHASE_DECL;
  // ---- global reads with clamped addresses (no branches): (a) the 4 x 6 x 16 x 16ch input tile ([n][2 groups][hi, lo]
  // [D/2][H/2][W/2] slots: 2 x 2 x 384 slots, three 16-byte copies per thread) and the weight fragments
  u32x4 pre[3], wpre[3];
  constexpr int NF = (16 * C9::VZ * C9::VY * 16 + NT - 1) / NT;      // F32: floats per thread of the 16 x 4 x 6 x 16 tile (column 16 is idle)
  constexpr int NWF = C9::WF32 / 4 / NT + 1;                        // F32: 16-byte pieces of the weight image per thread
  float pref[F32 ? NF : 1];
  u32x4 wpref[F32 ? NWF : 1];
  if constexpr (F32) {
    // rows of 16 floats: row R = (ci * VZ + vz) * VY + vy, 32 rows per pass of the workgroup
    const char* const src = reinterpret_cast<const char*>(p.u8 + (size_t)n * 16 * in_plane);
    const int vx = tid & 15;
    const int gx = ix0 + vx;
    const int xc = min(max(gx, 0), W2 - 1);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const unsigned R = (unsigned)(tid >> 4) + 32u * i;                 // < 16 * 24 = 384
      const unsigned ci = v3d::small_div<C9::VZ * C9::VY>(R), rem = R - ci * (C9::VZ * C9::VY);
      const unsigned uz = v3d::small_div<C9::VY>(rem);
      const int gz = iz0 + (int)uz, gy = iy0 + (int)(rem - uz * C9::VY);
      const bool ok = gz >= 0 && gz < D2 && gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
      const int zc = min(max(gz, 0), D2 - 1), yc = min(max(gy, 0), H2 - 1);
      const unsigned idx = __umul24(ci, (unsigned)in_plane) + __umul24(__umul24(zc, H2) + yc, W2) + xc;
      const float val = *reinterpret_cast<const float*>(src + idx * 4u);
      pref[i] = ok ? val : 0.f;
    }
    const u32x4* wq = reinterpret_cast<const u32x4*>(p.wbf);
#pragma unroll
    for (int i = 0; i < NWF; ++i) wpref[i] = wq[min(tid + NT * i, C9::WF32 / 4 - 1)];
  } else {
    const char* const src = reinterpret_cast<const char*>(reinterpret_cast<const u32x4*>(p.u8) + (size_t)n * 4 * in_plane);
    const int vx = tid & 15;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // slot it = tid + NT * i of the [4 (group, part)][VZ][VY][16] tile: row R = it / 16 = (gp * VZ + vz) * VY + vy
      const unsigned R = (unsigned)(tid >> 4) + 32u * i;                // 32 i .. 32 i + 31
      const bool up = R >= 24u * (i + 1);                                // gp = R / 24 = i or i + 1
      const unsigned rem = R - (up ? 24u * (i + 1) : 24u * i);           // R % 24
      const unsigned uz = __umul24(rem, 43u) >> 8;                       // rem / 6 (exact below 24)
      const int vz = (int)uz, vy = (int)(rem - __umul24(uz, 6u));
      const int gz = iz0 + vz, gy = iy0 + vy, gx = ix0 + vx;
      const bool ok = gz >= 0 && gz < D2 && gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
      const int zc = min(max(gz, 0), D2 - 1), yc = min(max(gy, 0), H2 - 1), xc = min(max(gx, 0), W2 - 1);
      const unsigned idx = (unsigned)in_plane * i + (up ? (unsigned)in_plane : 0u) + __umul24(__umul24(zc, H2) + yc, W2) + xc;   // < 2^28 (host check)
      const u32x4 val = V3D_C9_ABLATE == 5 ? (u32x4){1u, 2u, 3u, (unsigned)tid} : *reinterpret_cast<const u32x4*>(src + idx * 16u);
      pre[i] = ok ? val : (u32x4){0u, 0u, 0u, 0u};
    }
    const u32x4* wq = reinterpret_cast<const u32x4*>(p.wbf);
#pragma unroll
    for (int i = 0; i < 3; ++i) wpre[i] = wq[min(tid + NT * i, C9::WU32 / 4 - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
  // (b) the conv0 skip values of this lane's outputs, in flight during staging and the MFMA phase.  conv0's output is in
  // the split channel-last layout ([n][hi, lo][D][H][W] slots of 8 channels): channels cbase..cbase+3 are 8 bytes of the hi
  // slot and 8 bytes of the lo slot
  const int px = kq >> 1, cbase = 4 * (kq & 1);
  u32x2 skh[NCBI][4], skl[NCBI][4];
  float skf[F32 ? NCBI : 1][4][4];
  if constexpr (F32) {
    // fp32 [n, 8, D, H, W]: the lane's four channels are four planes apart
    const char* const skip = reinterpret_cast<const char*>(p.c0 + ((size_t)n * 8 + cbase) * out_plane);
    const int gx = ox0 - 1 + 2 * jn + px;
    const unsigned lofs = (unsigned)min(max(gx, 0), p.W - 1) * 4u;
    const unsigned cpl = (unsigned)out_plane * 4u;                       // bytes per channel plane (< 2^28, host check)
#pragma unroll
    for (int cbi = 0; cbi < NCBI; ++cbi) {
      const int cb = min(wave + 8 * cbi, C9::NCB - 1);
      const int cz = cb / C9::CY, cy = cb % C9::CY;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int gz = oz0 - 1 + 2 * cz + (rb >> 1), gy = oy0 - 1 + 2 * cy + (rb & 1);
        const int zc = min(max(gz, 0), p.D - 1), yc = min(max(gy, 0), p.H - 1);
        const unsigned rofs = (unsigned)((zc * p.H + yc) * p.W) * 4u;      // wave-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r) skf[cbi][rb][r] = *reinterpret_cast<const float*>(skip + (size_t)r * cpl + rofs + lofs);
      }
    }
  } else {
    const char* const skip = reinterpret_cast<const char*>(reinterpret_cast<const u32x2*>(p.c0) + ((size_t)n * 2 * out_plane) * 2);
    // the row (z, y) of a (cell row, rb) is wave-uniform: its base stays in SGPRs, the lane adds its x slot and channel half
    const int gx = ox0 - 1 + 2 * jn + px;
    const unsigned lofs = (unsigned)min(max(gx, 0), p.W - 1) * 16u + (unsigned)(kq & 1) * 8u;
    const unsigned lo_plane = (unsigned)out_plane * 16u;                  // hi slots -> lo slots, bytes (< 2^31, host check)
#pragma unroll
    for (int cbi = 0; cbi < NCBI; ++cbi) {
      const int cb = min(wave + 8 * cbi, C9::NCB - 1);
      const int cz = cb / C9::CY, cy = cb % C9::CY;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int gz = oz0 - 1 + 2 * cz + (rb >> 1), gy = oy0 - 1 + 2 * cy + (rb & 1);
        const int zc = min(max(gz, 0), p.D - 1), yc = min(max(gy, 0), p.H - 1);
        const unsigned rofs = (unsigned)((zc * p.H + yc) * p.W) * 16u;     // wave-uniform
        skh[cbi][rb] = V3D_C9_ABLATE == 4 ? (u32x2){rofs + lofs, 1u} : *reinterpret_cast<const u32x2*>(skip + rofs + lofs);
        skl[cbi][rb] = V3D_C9_ABLATE == 4 ? (u32x2){rofs + lofs, 2u} : *reinterpret_cast<const u32x2*>(skip + lo_plane + rofs + lofs);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- stage the input tile (slot = voxel * 2 + channel half (group), hi and lo arrays) and the weight fragments ------
  if constexpr (F32) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const unsigned R = (unsigned)(tid >> 4) + 32u * i;
      const unsigned ci = v3d::small_div<C9::VZ * C9::VY>(R), rem = R - ci * (C9::VZ * C9::VY);      // rem = vz * VY + vy
      if (R < 16 * C9::VZ * C9::VY) xf[ci * C9::FS + rem * C9::VX + (tid & 15)] = pref[i];
    }
#pragma unroll
    for (int i = 0; i < NWF; ++i)
      if (tid + NT * i < C9::WF32 / 4) reinterpret_cast<u32x4*>(wff)[tid + NT * i] = wpref[i];
    // the idle 17th voxel of every row is read by column 15: keep it finite
    for (int i = tid; i < 16 * C9::VZ * C9::VY; i += NT) {
      const unsigned ci = v3d::small_div<C9::VZ * C9::VY>((unsigned)i), rem = (unsigned)i - ci * (C9::VZ * C9::VY);
      xf[ci * C9::FS + rem * C9::VX + 16] = 0.f;
    }
  } else {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const unsigned R = (unsigned)(tid >> 4) + 32u * i;                   // as above
    const bool up = R >= 24u * (i + 1);
    const int gp = i + (up ? 1 : 0);
    const unsigned rem = R - (up ? 24u * (i + 1) : 24u * i);             // vz * VY + vy
    const int slot = (int)(__umul24(rem, (unsigned)C9::VX) + (unsigned)(tid & 15)) * 2 + (gp >> 1);
    ((gp & 1) ? xl : xh)[slot] = pre[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (tid + NT * i < C9::WU32 / 4) wfr[tid + NT * i] = wpre[i];
  if (tid < C9::VZ * C9::VY * 2) {         // the idle 17th voxel of every row is read by column 15: keep it finite
    const int slot = ((tid >> 1) * C9::VX + 16) * 2 + (tid & 1);
    xh[slot] = (u32x4){0u, 0u, 0u, 0u};
    xl[slot] = (u32x4){0u, 0u, 0u, 0u};
  }
  }
  PHASE_MARK(0);
  __syncthreads();
  PHASE_MARK(1);

  // ---- deconvolution: wave w owns cell rows w and w + 8; weight block (tz3, ty3), tz3 = {(pz 0, dz 0), (0, 1), (1, 1)},
  // likewise ty3 -------------------------------------------------------------------------------------------------------
  f32x4 acc[NCBI][4];
#pragma unroll
  for (int i = 0; i < NCBI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cbi = 0; cbi < NCBI; ++cbi) {
    const int cb = wave + 8 * cbi;
    if (cb < C9::NCB && V3D_C9_ABLATE != 3) {
      const int cz = cb / C9::CY, cy = cb % C9::CY;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          if constexpr (F32) {
            // k slice s, lane group kq: k = 4 s + kq = dx * 16 + ci  =>  dx = s >> 2, ci = 4 (s & 3) + kq
            float bv[8];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
              bv[sl] = xf[(4 * (sl & 3) + kq) * C9::FS + ((cz + dz) * C9::VY + (cy + dy)) * C9::VX + jn + (sl >> 2)];
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                if ((pz == 0 || dz == 1) && (py == 0 || dy == 1)) {
                  const int blk = (pz == 1 ? 2 : dz) * 3 + (py == 1 ? 2 : dy);
                  const f32x4 a0 = *reinterpret_cast<const f32x4*>(wff + ((blk * 2 + 0) * 64 + lane) * 4);
                  const f32x4 a1 = *reinterpret_cast<const f32x4*>(wff + ((blk * 2 + 1) * 64 + lane) * 4);
                  f32x4& c = acc[cbi][pz * 2 + py];
#pragma unroll
                  for (int sl = 0; sl < 8; ++sl)
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(sl < 4 ? a0[sl] : a1[sl - 4], bv[sl], c, 0, 0, 0);
                }
              }
            continue;
          }
          const int slot = (((cz + dz) * C9::VY + (cy + dy)) * C9::VX + jn + (kq >> 1)) * 2 + (kq & 1);
          const bf16x8 b_hi = __builtin_bit_cast(bf16x8, xh[slot]);
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, xl[slot]);
#pragma unroll
          for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              if ((pz == 0 || dz == 1) && (py == 0 || dy == 1)) {
                const int blk = (pz == 1 ? 2 : dz) * 3 + (py == 1 ? 2 : dy);
                const bf16x8 a_hi = __builtin_bit_cast(bf16x8, wfr[(blk * 2) * 64 + lane]);
                const bf16x8 a_lo = __builtin_bit_cast(bf16x8, wfr[(blk * 2 + 1) * 64 + lane]);
                f32x4& c = acc[cbi][pz * 2 + py];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, c, 0, 0, 0);
              }
            }
        }
    }
  }
  PHASE_MARK(2);
  __syncthreads();      // the input tile and the weight fragments are dead: their LDS becomes the u9 tile
  PHASE_MARK(3);

  // ---- BN bias + ReLU + conv0 skip -> u9 tile (zero outside the volume = the prob conv's padding) ------------
  // u9 tile layout [4 channel pairs][HD][HH][RS x][2]: a lane's channels (r, r+1) are one 8-byte write, and the
  // prob conv below multiplies both channels of a pair with one packed FMA.
  {
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = p.bias9[cbase + r];
#pragma unroll
    for (int cbi = 0; cbi < NCBI; ++cbi) {
      const int cb = wave + 8 * cbi;
      if (cb < C9::NCB && jn < C9::CX) {
        const int cz = cb / C9::CY, cy = cb % C9::CY;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const int hz = 2 * cz + (rb >> 1), hy = 2 * cy + (rb & 1), hx = 2 * jn + px;
          const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
          const bool inside = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          float skv[4];
          if constexpr (F32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) skv[r] = skf[cbi][rb][r];
          } else {
            const u32x2 sh = skh[cbi][rb], sl = skl[cbi][rb];
            skv[0] = __uint_as_float(sh.x << 16) + __uint_as_float(sl.x << 16);
            skv[1] = __uint_as_float(sh.x & 0xffff0000u) + __uint_as_float(sl.x & 0xffff0000u);
            skv[2] = __uint_as_float(sh.y << 16) + __uint_as_float(sl.y << 16);
            skv[3] = __uint_as_float(sh.y & 0xffff0000u) + __uint_as_float(sl.y & 0xffff0000u);
          }
          float val[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) val[r] = inside ? fmaxf(acc[cbi][rb][r] + bias[r], 0.f) + skv[r] : 0.f;
#pragma unroll
          for (int rp = 0; rp < 2; ++rp)
            *reinterpret_cast<f32x2*>(u9s + ((((cbase >> 1) + rp) * (C9::HD * C9::HH) + hz * C9::HH + hy) * C9::RS + hx) * 2) =
                (f32x2){val[2 * rp], val[2 * rp + 1]};
        }
      }
    }
  }
  PHASE_MARK(4);
  __syncthreads();
  PHASE_MARK(5);

  // ---- prob conv: wave = (z plane, y half), lane = (y of 4, 2 consecutive x); both channels of a pair per packed FMA --
  // (a register-ring software pipeline of the LDS reads over the 12 (pair, kz) stages measured the same 0.45 ms: this
  // phase costs 0.11 ms of the kernel, the rest is the chain loads -> staging -> MFMA -> u9 tile)
  {
    const int z = wave >> 1, y = (wave & 1) * 4 + (lane >> 4), xp = lane & 15;
    if (xp < C9::TW / 2) {
      // six accumulation chains (one pair per kz, added at the end) instead of two: a dependent v_pk_fma_f32 costs ~20 cycles
      // (measured on the depth-march experiment, conv9z.hip, which sums in the same order)
      f32x2 o0[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, o1[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      const f32x2* wp2 = reinterpret_cast<const f32x2*>(p.wprob);       // [4 pairs][27 taps][2]
#pragma unroll 1
      for (int cp = 0; cp < (V3D_C9_ABLATE == 1 ? 0 : V3D_C9_ABLATE == 2 ? 1 : 4); ++cp) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kz = 0; kz < 3; ++kz) {
            const f32x2* row = reinterpret_cast<const f32x2*>(u9s) +
                               (cp * (C9::HD * C9::HH) + (z + kz) * C9::HH + (y + ky)) * C9::RS + 2 * xp;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(row), q1 = *reinterpret_cast<const f32x4*>(row + 2);
            const f32x2 a0 = {q0.x, q0.y}, a1 = {q0.z, q0.w}, a2 = {q1.x, q1.y}, a3 = {q1.z, q1.w};
            const f32x2* wk = wp2 + (cp * 27 + (kz * 3 + ky) * 3);           // wave-uniform -> s_load
            const f32x2 w0 = wk[0], w1 = wk[1], w2 = wk[2];
            o0[kz] += a0 * w0; o1[kz] += a1 * w0;
            o0[kz] += a1 * w1; o1[kz] += a2 * w1;
            o0[kz] += a2 * w2; o1[kz] += a3 * w2;
          }
        }
      }
      const f32x2 s0 = (o0[0] + o0[1]) + o0[2], s1 = (o1[0] + o1[1]) + o1[2];
      const int gz = oz0 + z, gy = oy0 + y, gx = ox0 + 2 * xp;
      if (gz < p.D && gy < p.H && gx < p.W) {
        const float bsv = p.bprob[0];
        const f32x2 res = {s0.x + s0.y + bsv, s1.x + s1.y + bsv};
        float* o = p.out + ((size_t)n * out_plane + (size_t)(gz * p.H) * p.W) + (unsigned)(__umul24(gy, p.W) + gx);   // uniform base + lane offset
        if (gx + 1 < p.W && (p.W & 1) == 0) {
          *reinterpret_cast<f32x2*>(o) = res;
        } else {
          o[0] = res.x;
          if (gx + 1 < p.W) o[1] = res.y;
        }
      }
    }
  }
  PHASE_MARK(6);
  PHASE_FLUSH;
}

// ---- fp32 [n, C, D, H, W] -> split channel-last layout (per-layer entry point of the split kernels) ----------------
__global__ __launch_bounds__(256) void encode_split_kernel(const float* __restrict__ in, u32x4* __restrict__ out, int C,
                                                           size_t plane, size_t total) {   // total = n * (C / 8) * plane
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t sp = i % plane, ng = i / plane;                 // ng = n * (C / 8) + group
  const float* src = in + (ng * 8) * plane + sp;               // channels of a group are consecutive planes
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = src[(size_t)c * plane];
  u32x2 ha, la, hb, lb;
  split4(v[0], v[1], v[2], v[3], ha, la);
  split4(v[4], v[5], v[6], v[7], hb, lb);
  out[(ng * 2) * plane + sp] = (u32x4){ha.x, ha.y, hb.x, hb.y};
  out[(ng * 2 + 1) * plane + sp] = (u32x4){la.x, la.y, lb.x, lb.y};
}

// ---- soft-argmin over D (mvsnet.py:219-227): p = softmax(-x), depth = sum_d vals[d] p[d] ----------
__global__ __launch_bounds__(256) void soft_argmin_kernel(const float* __restrict__ reg,
                                                          const float* __restrict__ vals,
                                                          float* __restrict__ depth, int n, int D,
                                                          int HW) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)n * HW) return;
  const int b = gid / HW, pix = gid % HW;
  const float* col = reg + (size_t)b * D * HW + pix;
  // one pass over D with a running maximum: num = sum vals_d e^{-x_d - m}, den = sum e^{-x_d - m}
  float m = -INFINITY, num = 0.f, den = 0.f;
  auto step = [&](float xin, float val) __attribute__((always_inline)) {
    const float x = -xin;
    if (x > m) {
      const float sc = expf(m - x);          // exp(-inf) = 0 on the first plane
      num *= sc; den *= sc; m = x;
    }
    const float ex = expf(x - m);
    num += val * ex;
    den += ex;
  };
  // 8 planes are requested before the first is consumed (the running-maximum chain is sequential, the loads are not)
  int d = 0;
  for (; d + 8 <= D; d += 8) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_nontemporal_load(col + (size_t)(d + i) * HW);
#pragma unroll
    for (int i = 0; i < 8; ++i) step(x[i], vals[d + i]);
  }
  for (; d < D; ++d) step(col[(size_t)d * HW], vals[d]);
  const float e = num / den;
  depth[gid] = e;
}

// ---- layer table ---------------------------------------------------------------------------------
//                       mode      Cin Cout  TD TH  TW  CK
#ifndef V3D_L0_CFG
#define V3D_L0_CFG kConvS1Pair, 32, 8, 4, 8, 28, 4, 3
#endif
typedef ConvCfg<V3D_L0_CFG> L0;
// CostRegNet(16, 8) (the reference's signature default feat_dim = 16, lightningmodel.py:18): the exact-fp32 conv0 with 16 input
// channels; the split-bf16 / depth-march conv0 kernels run on the volume zero-extended to 32 channels with zero weights
typedef ConvCfg<kConvS1Pair, 16, 8, 4, 8, 28, 4, 3> L0_16;
#ifndef V3D_L1_CFG
#define V3D_L1_CFG kConvS2, 8, 16, 2, 4, 28, 4
#endif
typedef ConvCfg<V3D_L1_CFG> L1;
#ifndef V3D_L2_CFG
#define V3D_L2_CFG kConvS1, 16, 16, 4, 4, 28, 8
#endif
typedef ConvCfg<V3D_L2_CFG> L2;
typedef ConvCfg<kConvS2, 16, 32, 2, 7, 14, 4> L3;
typedef ConvCfg<kConvS1, 32, 32, 4, 7, 14, 8> L4;
typedef ConvCfg<kConvS2, 32, 64, 2, 4, 7, 8> L5;
typedef ConvCfg<kConvS1, 64, 64, 2, 4, 7, 16> L6;
typedef ConvCfg<kDeconvS2, 64, 32, 4, 8, 14, 16> L7;
#ifndef V3D_L8_CFG
#define V3D_L8_CFG kDeconvS2, 32, 16, 4, 8, 28, 8
#endif
typedef ConvCfg<V3D_L8_CFG> L8;
#ifndef V3D_L9_CFG
#define V3D_L9_CFG kDeconvS2, 16, 8, 4, 8, 28, 16, 3
#endif
typedef ConvCfg<V3D_L9_CFG> L9;

struct LayerDesc { int mode, cin, cout, ck; };
const LayerDesc kLayers[10] = {
    {L0::MODE, 32, 8, L0::CK},   {L1::MODE, 8, 16, L1::CK},   {L2::MODE, 16, 16, L2::CK},
    {kConvS2, 16, 32, L3::CK},  {kConvS1, 32, 32, L4::CK},  {kConvS2, 32, 64, L5::CK},
    {kConvS1, 64, 64, L6::CK},  {kDeconvS2, 64, 32, L7::CK}, {kDeconvS2, 32, 16, L8::CK},
    {kDeconvS2, 16, 8, L9::CK}};

template <class C>
int launch_conv(const char* name, const float* in, const float* wp, const float* bias,
                const float* skip, float* out, int n, int Di, int Hi, int Wi, hipStream_t s) {
  ConvParams p;
  p.in = in; p.wp = wp; p.bias = bias; p.skip = skip; p.out = out; p.n = n;
  p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  if (C::S1LIKE) { p.Do = Di; p.Ho = Hi; p.Wo = Wi; }
  else if (C::MODE == kConvS2) { p.Do = (Di - 1) / 2 + 1; p.Ho = (Hi - 1) / 2 + 1; p.Wo = (Wi - 1) / 2 + 1; }
  else { p.Do = 2 * Di; p.Ho = 2 * Hi; p.Wo = 2 * Wi; }
  p.ntz = (p.Do + C::TD - 1) / C::TD; p.nty = (p.Ho + C::TH - 1) / C::TH; p.ntx = (p.Wo + C::TW - 1) / C::TW;
  p.relu = 1;
  p.zy_order = 0;
  const long long blocks = (long long)n * p.ntz * p.nty * p.ntx;
  V3D_REQUIRE(blocks > 0 && blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv3d: bad grid");
  V3D_REQUIRE((long long)C::CK * Di * Hi * Wi < (1ll << 31), V3D_ERR_BAD_SHAPE,
              "conv3d: input volume too large for 32-bit tile offsets");
  {
    v3d::TimedScope ts(name, s);
    // (conv0 only: on conv2 the 40 extra staging registers cost more than the loads save: 0.37 -> 0.41 ms)
    constexpr bool kVecOk = C::MODE == kConvS1Pair && C::PF && C::TW % 4 == 0 && C::IW <= 61;
    const bool vec = kVecOk && Wi % 4 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0 && v3d::option(v3d::kOptConvVec) != 0;
    if constexpr (kVecOk) {
      if (vec) conv3d_mfma_kernel<C, true><<<(unsigned)blocks, 256, 0, s>>>(p);
      else conv3d_mfma_kernel<C, false><<<(unsigned)blocks, 256, 0, s>>>(p);
    } else {
      conv3d_mfma_kernel<C, false><<<(unsigned)blocks, 256, 0, s>>>(p);
    }
  }
  V3D_CHECK_LAUNCH("conv3d_mfma_kernel");
  return V3D_OK;
}

}  // namespace

namespace {
#ifndef V3D_C0_WAVES
#define V3D_C0_WAVES 8
#endif
constexpr int kC0Waves = V3D_C0_WAVES;      // waves per workgroup of the fused path's conv0 (4 or 8)
int launch_conv0_bf16(bool split_in, bool split_out, const float* in, const float* wbf, const float* bias, const float* skip, float* out, int n,
                      int Di, int Hi, int Wi, hipStream_t s) {
  ConvParams p;
  p.in = in; p.wp = wbf; p.bias = bias; p.skip = skip; p.out = out; p.n = n;
  p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Do = Di; p.Ho = Hi; p.Wo = Wi;
  p.ntz = (Di + C0::TD - 1) / C0::TD; p.nty = (Hi + C0::TH - 1) / C0::TH; p.ntx = (Wi + C0::TW - 1) / C0::TW;
  p.relu = 1;
  p.zy_order = tile_order(p.ntx, p.nty);
  const long long blocks = (long long)n * p.ntz * p.nty * p.ntx;
  V3D_REQUIRE(blocks > 0 && blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv0: bad grid");
  V3D_REQUIRE((long long)8 * Di * Hi * Wi < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv0: input volume too large");
  static bool attr_set[64] = {false};      // per device: the dynamic-LDS opt-in is a per-device function attribute
  int attr_dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&attr_dev));
  V3D_REQUIRE(attr_dev >= 0 && attr_dev < 64, V3D_ERR_UNSUPPORTED, "device ordinal %d", attr_dev);
  if (!attr_set[attr_dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv0_bf16x2_kernel<false, false, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C0::LDS_BYTES));
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv0_bf16x2_kernel<true, false, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C0::LDS_BYTES));
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv0_bf16x2_kernel<true, true, kC0Waves>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C0::LDS_BYTES));
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv0_bf16x2_kernel<false, true, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C0::LDS_BYTES));
    attr_set[attr_dev] = true;
  }
  {
    v3d::TimedScope ts("costreg_conv0", s);
#ifdef V3D_C0_PERSIST       // developer A/B: 512 resident workgroups walking their tiles (same time, 15 % more halo traffic)
    const unsigned grid = v3d::persistent_grid(blocks, 2);        // 76.8 KB of LDS: two workgroups per CU
#else
    const unsigned grid = (unsigned)((blocks + 7) / 8 * 8);       // one tile per workgroup: the walk has a single step
#endif
    if (split_in && split_out) conv0_bf16x2_kernel<true, true, kC0Waves><<<grid, 64 * kC0Waves, C0::LDS_BYTES, s>>>(p);
    else if (split_in) conv0_bf16x2_kernel<true, false, 4><<<grid, 256, C0::LDS_BYTES, s>>>(p);
    else if (split_out) conv0_bf16x2_kernel<false, true, 4><<<grid, 256, C0::LDS_BYTES, s>>>(p);
    else conv0_bf16x2_kernel<false, false, 4><<<grid, 256, C0::LDS_BYTES, s>>>(p);
  }
  V3D_CHECK_LAUNCH("conv0_bf16x2_kernel");
  return V3D_OK;
}
}  // namespace

namespace {
template <class C>
int launch_convg(const char* name, const void* in, const float* wbf, const float* bias, float* out_f32, void* out_split,
                 int n, int Di, int Hi, int Wi, hipStream_t s) {
  ConvGParams p;
  p.in = in; p.wp = wbf; p.bias = bias; p.out_f32 = out_f32; p.out_split = out_split; p.n = n;
  p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.Do = (Di - 1) / C::S + 1; p.Ho = (Hi - 1) / C::S + 1; p.Wo = (Wi - 1) / C::S + 1;
  p.ntz = (p.Do + C::TD - 1) / C::TD; p.nty = (p.Ho + C::TH - 1) / C::TH; p.ntx = (p.Wo + C::TW - 1) / C::TW;
  const long long blocks = (long long)n * p.ntz * p.nty * p.ntx;
  V3D_REQUIRE(blocks > 0 && blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "%s: bad grid", name);
  V3D_REQUIRE(((C::OUT & kOutF32) == 0 || out_f32) && ((C::OUT & kOutSplit) == 0 || out_split), V3D_ERR_BAD_ARG,
              "%s: missing output buffer", name);
  p.m_tx = v3d::magic_u32((unsigned long long)blocks, (unsigned)p.ntx);
  p.m_ty = v3d::magic_u32((unsigned long long)blocks / p.ntx + 1, (unsigned)p.nty);
  p.m_tz = v3d::magic_u32((unsigned long long)blocks / p.ntx / p.nty + 1, (unsigned)p.ntz);
  // 32-bit slot offsets inside one (view, channel group): 24-bit multiplies on the axes, 2 (hi, lo) x plane x 16 bytes < 4 GB;
  // the output side indexes 4 planes from the item's base
  V3D_REQUIRE(Di < (1 << 12) && Hi < (1 << 12) && Wi < (1 << 12) && (long long)Di * Hi * Wi < (1ll << 24), V3D_ERR_BAD_SHAPE,
              "%s: volume %d x %d x %d too large for 32-bit slot offsets", name, Di, Hi, Wi);
  static bool attr_set[64] = {false};      // per device: the dynamic-LDS opt-in is a per-device function attribute
  int attr_dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&attr_dev));
  V3D_REQUIRE(attr_dev >= 0 && attr_dev < 64, V3D_ERR_UNSUPPORTED, "device ordinal %d", attr_dev);
  if (!attr_set[attr_dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)convg_bf16x2_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C::LDS_BYTES));
    attr_set[attr_dev] = true;
  }
  {
    v3d::TimedScope ts(name, s);
    unsigned grid = (unsigned)((blocks * C::NCG + 7) / 8 * 8);        // one (tile, channel group) item per workgroup ...
    if (C::PERSIST) {                                                  // ... or as many workgroups as the chip holds
      static int wgs_per_cu = 0;
      if (wgs_per_cu == 0) {
        int nb = 0;
        V3D_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)convg_bf16x2_kernel<C>, 256, C::LDS_BYTES));
        wgs_per_cu = nb > 0 ? nb : 1;
      }
      grid = v3d::persistent_grid(blocks * C::NCG, wgs_per_cu);
    }
    convg_bf16x2_kernel<C><<<grid, 256, C::LDS_BYTES, s>>>(p, p.bias);
  }
  V3D_CHECK_LAUNCH(name);
  return V3D_OK;
}
}  // namespace

namespace {
template <class C>
int launch_deconvg(const char* name, const void* in, const float* wbf, const float* bias, const float* skip, float* out_f32,
                   void* out_split, int n, int Di, int Hi, int Wi, hipStream_t s) {
  DeconvGParams p;
  p.in = in; p.wp = wbf; p.bias = bias; p.skip = skip; p.out_f32 = out_f32; p.out_split = out_split; p.n = n;
  p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.ntz = (Di + C::CZ - 1) / C::CZ; p.nty = (Hi + C::CY - 1) / C::CY; p.ntx = (Wi + C::CX - 1) / C::CX;
  const long long blocks = (long long)n * p.ntz * p.nty * p.ntx;
  V3D_REQUIRE(blocks > 0 && blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "%s: bad grid", name);
  V3D_REQUIRE(skip && ((C::OUT & kOutF32) == 0 || out_f32) && ((C::OUT & kOutSplit) == 0 || out_split), V3D_ERR_BAD_ARG,
              "%s: missing buffer", name);
  V3D_REQUIRE(Di < (1 << 12) && Hi < (1 << 12) && Wi < (1 << 12) && (long long)Di * Hi * Wi < (1ll << 24), V3D_ERR_BAD_SHAPE,
              "%s: volume %d x %d x %d too large for 32-bit slot offsets", name, Di, Hi, Wi);
  p.m_tx = v3d::magic_u32((unsigned long long)blocks, (unsigned)p.ntx);
  p.m_ty = v3d::magic_u32((unsigned long long)blocks / p.ntx + 1, (unsigned)p.nty);
  p.m_tz = v3d::magic_u32((unsigned long long)blocks / p.ntx / p.nty + 1, (unsigned)p.ntz);
  {
    v3d::TimedScope ts(name, s);
    deconvg_bf16x2_kernel<C><<<dim3((unsigned)blocks, C::NCG), 256, 0, s>>>(p);
  }
  V3D_CHECK_LAUNCH(name);
  return V3D_OK;
}
}  // namespace

struct v3d_costreg_weights {
  int in_channels, base;
  float* dev;                 // one allocation holding everything below
  size_t wp_ofs[10], bias_ofs[10], prob_w_ofs, prob_w2_ofs, prob_b_ofs, c0bf_ofs, c0f32_ofs, cgbf_ofs[7], dgbf_ofs[2], c9bf_ofs, c9f32_ofs, total;
};

extern "C" int v3d_costreg_pack(const float* const* conv_w, const float* const* bn_w,
                                const float* const* bn_b, const float* const* bn_m,
                                const float* const* bn_v, const float* prob_w, const float* prob_b,
                                int in_channels, int base, float eps,
                                v3d_costreg_weights** out_handle) {
  V3D_REQUIRE(conv_w && bn_w && bn_b && bn_m && bn_v && prob_w && prob_b && out_handle,
              V3D_ERR_BAD_ARG, "v3d_costreg_pack: null argument");
  V3D_REQUIRE((in_channels == 32 || in_channels == 16) && base == 8, V3D_ERR_UNSUPPORTED,
              "v3d_costreg_pack: CostRegNet(32, 8) and CostRegNet(16, 8) are built (got %d, %d)", in_channels, base);
  v3d_costreg_weights* h = new v3d_costreg_weights();
  h->in_channels = in_channels; h->base = base;
  std::vector<float> host;
  auto reserve = [&](size_t nfloat) { size_t o = host.size(); host.resize(o + (nfloat + 63) / 64 * 64, 0.f); return o; };
  for (int l = 0; l < 10; ++l) {
    LayerDesc L = kLayers[l];
    if (l == 0) L.cin = in_channels;
    const bool pair = L.mode == kConvS1Pair;
    const int MB = pair ? 1 : (L.cout + 15) / 16, C4 = L.ck / 4, nchunk = L.cin / L.ck;
    const int KXN = pair ? 4 : 3, NT = 9 * KXN;
    h->wp_ofs[l] = reserve((size_t)nchunk * NT * C4 * MB * 64);
    h->bias_ofs[l] = reserve(L.cout);
    float* wp = host.data() + h->wp_ofs[l];
    float* bias = host.data() + h->bias_ofs[l];
    std::vector<float> scale(L.cout);
    for (int co = 0; co < L.cout; ++co) {
      // eval-mode BatchNorm: y = (x - mean) / sqrt(var + eps) * gamma + beta   (mvsnet.py:22,33)
      scale[co] = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
      bias[co] = bn_b[l][co] - bn_m[l][co] * scale[co];
    }
    for (int chunk = 0; chunk < nchunk; ++chunk)
      for (int tap = 0; tap < NT; ++tap)
        for (int c4 = 0; c4 < C4; ++c4)
          for (int m = 0; m < MB; ++m)
            for (int lane = 0; lane < 64; ++lane) {
              const int row = m * 16 + (lane & 15);
              const int ci = chunk * L.ck + c4 * 4 + (lane >> 4);
              float v = 0.f;
              if (pair) {
                // row = x-shift * 8 + channel; virtual tap = (kz, ky, kx') with kx' in 0..3
                const int sx = row >> 3, co = row & 7, kzy = tap / 4, kx = tap % 4 - sx;
                if (kx >= 0 && kx <= 2)
                  v = conv_w[l][((size_t)co * L.cin + ci) * 27 + kzy * 3 + kx] * scale[co];
              } else if (row < L.cout) {
                // Conv3d weight [Co, Ci, 3,3,3]; ConvTranspose3d weight [Ci, Co, 3,3,3]
                const size_t idx = L.mode == kDeconvS2 ? ((size_t)ci * L.cout + row) * 27 + tap
                                                       : ((size_t)row * L.cin + ci) * 27 + tap;
                v = conv_w[l][idx] * scale[row];
              }
              wp[((((size_t)chunk * NT + tap) * C4 + c4) * MB + m) * 64 + lane] = v;
            }
  }
  {
    // split-bf16 image of conv0 for conv0_bf16x2_kernel: [chunk 4][kzy 9][hi, lo][lane 64][4 words]
    const int l = 0;
    h->c0bf_ofs = reserve((size_t)C0::NCH * C0::WU32);
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->c0bf_ofs);
    auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int chunk = 0; chunk < C0::NCH; ++chunk)
      for (int kzy = 0; kzy < 9; ++kzy)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = lane & 15, kxp = lane >> 4, sx = row >> 3, co = row & 7, kx = kxp - sx;
          unsigned hi[8], lo[8];
          for (int e = 0; e < 8; ++e) {
            const int ci = chunk * 8 + e;
            float v = 0.f;
            if (kx >= 0 && kx <= 2 && ci < in_channels) {        // (16 input channels: chunks 2, 3 carry zero weights)
              const float sc = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
              v = conv_w[l][((size_t)co * in_channels + ci) * 27 + kzy * 3 + kx] * sc;
            }
            hi[e] = rne(v);
            lo[e] = rne(v - up(hi[e]));
          }
          for (int part = 0; part < 2; ++part) {
            const unsigned* src = part ? lo : hi;
            unsigned* dst = wb + (((size_t)chunk * 9 + kzy) * 2 + part) * 256 + lane * 4;
            for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
          }
        }
  }
  {
    // exact-fp32 image of conv0 for conv0z_kernel<true> (v_mfma_f32_16x16x4_f32, pair mode): [chunk 4][kz * 3 + ky][channel 8]
    // [lane 64]; lane (kq, m): row m = x shift * 8 + co, k = x tap kq of the 4-wide window
    const int l = 0;
    h->c0f32_ofs = reserve((size_t)4 * 9 * 8 * 64);
    float* wf = host.data() + h->c0f32_ofs;
    for (int chunk = 0; chunk < 4; ++chunk)
      for (int kzy = 0; kzy < 9; ++kzy)
        for (int e = 0; e < 8; ++e)
          for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, kxp = lane >> 4, sx = row >> 3, co = row & 7, kx = kxp - sx, ci = chunk * 8 + e;
            float v = 0.f;
            if (kx >= 0 && kx <= 2 && ci < in_channels) {
              const float sc = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
              v = conv_w[l][((size_t)co * in_channels + ci) * 27 + kzy * 3 + kx] * sc;
            }
            wf[(((size_t)chunk * 9 + kzy) * 8 + e) * 64 + lane] = v;
          }
  }
  {
    // split-bf16 image of conv9 for conv9_prob_kernel: [block 9][hi, lo][lane 64][4 words]; block = tz3 * 3 + ty3 with
    // t?3 = {(parity 0, input 0), (0, 1), (1, 1)} <-> kernel tap {2, 0, 1}; rows = x parity * 8 + co, k = x input * 16 + ci
    const int l = 9;
    h->c9bf_ofs = reserve((size_t)C9::WU32);
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->c9bf_ofs);
    auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
    static const int tap3[3] = {2, 0, 1};
    for (int blk = 0; blk < 9; ++blk)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, px = m >> 3, co = m & 7, kq = lane >> 4, dx = kq >> 1;
        const int kz = tap3[blk / 3], ky = tap3[blk % 3];
        const int kx = px == 0 ? (dx == 0 ? 2 : 0) : (dx == 1 ? 1 : -1);
        const float sc = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
        unsigned hi[8], lo[8];
        for (int e = 0; e < 8; ++e) {
          const int ci = (kq & 1) * 8 + e;
          const float v = kx < 0 ? 0.f : conv_w[l][((size_t)ci * 8 + co) * 27 + kz * 9 + ky * 3 + kx] * sc;
          hi[e] = rne(v);
          lo[e] = rne(v - up(hi[e]));
        }
        for (int part = 0; part < 2; ++part) {
          const unsigned* src = part ? lo : hi;
          unsigned* dst = wb + ((size_t)blk * 2 + part) * 256 + lane * 4;
          for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
        }
      }
    // fp32 image of the same GEMM for conv9_prob_kernel<true>: [block 9][half 2][lane 64][4]; k slice sl = half * 4 + q of
    // v_mfma_f32_16x16x4_f32, lane group kq: k = 4 sl + kq = x input * 16 + ci
    h->c9f32_ofs = reserve((size_t)C9::WF32);
    float* wf = host.data() + h->c9f32_ofs;
    for (int blk = 0; blk < 9; ++blk)
      for (int sl = 0; sl < 8; ++sl)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, px = m >> 3, co = m & 7, kq = lane >> 4, k = 4 * sl + kq, dx = k >> 4, ci = k & 15;
          const int kz = tap3[blk / 3], ky = tap3[blk % 3];
          const int kx = px == 0 ? (dx == 0 ? 2 : 0) : (dx == 1 ? 1 : -1);
          const float sc = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
          wf[(((size_t)blk * 2 + (sl >> 2)) * 64 + lane) * 4 + (sl & 3)] =
              kx < 0 ? 0.f : conv_w[l][((size_t)ci * 8 + co) * 27 + kz * 9 + ky * 3 + kx] * sc;
        }
  }
  for (int l = 1; l <= 6; ++l) {
    // split-bf16 images of conv1..conv6 for convg_bf16x2_kernel: [cout group][8-channel chunk][kz, ky][hi, lo][lane 64]
    // [4 words]; rows = output channel within the group, k = 8 * x tap + ci (x tap 3 = 0)
    static const int cins[7] = {0, 8, 16, 16, 32, 32, 64}, couts[7] = {0, 16, 16, 32, 32, 64, 64};
    const int cin = cins[l], cout = couts[l], nch = cin / 8, ncg = cout / 16;
    const size_t words = (size_t)ncg * nch * 9 * 2 * 64 * 4;
    h->cgbf_ofs[l] = reserve(words);
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->cgbf_ofs[l]);
    auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int g = 0; g < ncg; ++g)
      for (int ch = 0; ch < nch; ++ch)
        for (int kzy = 0; kzy < 9; ++kzy)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = g * 16 + (lane & 15), kx = lane >> 4;
            const float sc = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
            unsigned hi[8], lo[8];
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 8 + e;
              const float v = kx > 2 ? 0.f : conv_w[l][((size_t)co * cin + ci) * 27 + kzy * 3 + kx] * sc;
              hi[e] = rne(v);
              lo[e] = rne(v - up(hi[e]));
            }
            for (int part = 0; part < 2; ++part) {
              const unsigned* src = part ? lo : hi;
              unsigned* dst = wb + ((((size_t)g * nch + ch) * 9 + kzy) * 2 + part) * 256 + lane * 4;
              for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
            }
          }
  }
  for (int l = 7; l <= 8; ++l) {
    // split-bf16 images of conv7 / conv8 for deconvg_bf16x2_kernel: [cout group][8-channel chunk][block 12][hi, lo]
    // [lane 64][4 words]; block = zb * 4 + (py, px), zb = {(pz 0, dz 0), (1, 0), (1, 1)}; rows = output channel,
    // k = 8 * (dy, dx) + ci.  Kernel tap of (parity p, input d): p 0: d 0 -> 1; p 1: d 0 -> 2, d 1 -> 0.
    const int cin = l == 7 ? 64 : 32, cout = l == 7 ? 32 : 16, nch = cin / 8, ncg = cout / 16;
    h->dgbf_ofs[l - 7] = reserve((size_t)ncg * nch * 12 * 2 * 64 * 4);
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->dgbf_ofs[l - 7]);
    auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
    auto tap = [](int par, int d) { return par == 0 ? (d == 0 ? 1 : -1) : (d == 0 ? 2 : 0); };
    for (int g = 0; g < ncg; ++g)
      for (int ch = 0; ch < nch; ++ch)
        for (int blk = 0; blk < 12; ++blk)
          for (int lane = 0; lane < 64; ++lane) {
            const int zb = blk / 4, py = (blk >> 1) & 1, px = blk & 1;
            const int pz = zb == 0 ? 0 : 1, dz = zb == 2 ? 1 : 0;
            const int co = g * 16 + (lane & 15), dy = lane >> 5, dx = (lane >> 4) & 1;
            const int kz = tap(pz, dz), ky = tap(py, dy), kx = tap(px, dx);
            const float sc = bn_w[l][co] / sqrtf(bn_v[l][co] + eps);
            unsigned hi[8], lo[8];
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 8 + e;
              const float v = (kz < 0 || ky < 0 || kx < 0)
                                  ? 0.f
                                  : conv_w[l][((size_t)ci * cout + co) * 27 + kz * 9 + ky * 3 + kx] * sc;
              hi[e] = rne(v);
              lo[e] = rne(v - up(hi[e]));
            }
            for (int part = 0; part < 2; ++part) {
              const unsigned* src = part ? lo : hi;
              unsigned* dst = wb + ((((size_t)g * nch + ch) * 12 + blk) * 2 + part) * 256 + lane * 4;
              for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
            }
          }
  }
  {
    // prob weights for the fused kernel, channel pairs interleaved: [4 pairs][27 taps][2]
    h->prob_w2_ofs = reserve((size_t)base * 27);
    for (int cp = 0; cp < 4; ++cp)
      for (int t = 0; t < 27; ++t)
        for (int e = 0; e < 2; ++e) host[h->prob_w2_ofs + ((size_t)cp * 27 + t) * 2 + e] = prob_w[(cp * 2 + e) * 27 + t];
  }
  h->prob_w_ofs = reserve((size_t)base * 27);
  memcpy(host.data() + h->prob_w_ofs, prob_w, sizeof(float) * base * 27);
  h->prob_b_ofs = reserve(1);
  host[h->prob_b_ofs] = prob_b[0];
  h->total = host.size();
  hipError_t e = hipMalloc((void**)&h->dev, h->total * sizeof(float));
  if (e != hipSuccess) { delete h; return v3d::fail(V3D_ERR_HIP, "hipMalloc(weights): %s", hipGetErrorString(e)); }
  e = hipMemcpy(h->dev, host.data(), h->total * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return v3d::fail(V3D_ERR_HIP, "hipMemcpy(weights): %s", hipGetErrorString(e)); }
  *out_handle = h;
  return V3D_OK;
}

extern "C" void v3d_costreg_free(v3d_costreg_weights* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  delete h;
}

static int run_layer(const v3d_costreg_weights* h, int layer, const float* in, const float* skip,
                     float* out, int n, int Di, int Hi, int Wi, int precision, hipStream_t s) {
  const float* wp = h->dev + h->wp_ofs[layer];
  const float* bias = h->dev + h->bias_ofs[layer];
  switch (layer) {
    case 0: {
      if (h->in_channels == 16) {
        V3D_REQUIRE(precision == V3D_PRECISION_FP32, V3D_ERR_UNSUPPORTED,
                    "costreg: the per-layer conv0 of CostRegNet(16, 8) is built for V3D_PRECISION_FP32 only (the chain entry points "
                    "cover both precisions)");
        return launch_conv<L0_16>("costreg_conv0", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
      }
      if (precision == V3D_PRECISION_FP32) return launch_conv<L0>("costreg_conv0", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
      return launch_conv0_bf16(false, false, in, h->dev + h->c0bf_ofs, bias, skip, out, n, Di, Hi, Wi, s);
    }
    case 1: return launch_conv<L1>("costreg_conv1", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 2: return launch_conv<L2>("costreg_conv2", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 3: return launch_conv<L3>("costreg_conv3", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 4: return launch_conv<L4>("costreg_conv4", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 5: return launch_conv<L5>("costreg_conv5", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 6: return launch_conv<L6>("costreg_conv6", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 7: return launch_conv<L7>("costreg_conv7", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 8: return launch_conv<L8>("costreg_conv8", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
    case 9: return launch_conv<L9>("costreg_conv9", in, wp, bias, skip, out, n, Di, Hi, Wi, s);
  }
  return v3d::fail(V3D_ERR_BAD_ARG, "costreg: layer %d out of range", layer);
}

extern "C" int v3d_costreg_layer_f32(const v3d_costreg_weights* h, int layer, const float* in,
                                     const float* skip, int n, int Di, int Hi, int Wi, float* out,
                                     int precision, void* stream) {
  V3D_REQUIRE(h && in && out, V3D_ERR_BAD_ARG, "v3d_costreg_layer_f32: null argument");
  V3D_REQUIRE(n > 0 && Di > 0 && Hi > 0 && Wi > 0, V3D_ERR_BAD_SHAPE, "v3d_costreg_layer_f32: bad shape");
  V3D_REQUIRE(precision == V3D_PRECISION_SPLIT_BF16 || precision == V3D_PRECISION_FP32, V3D_ERR_BAD_ARG,
              "v3d_costreg_layer_f32: unknown precision %d", precision);
  return run_layer(h, layer, in, skip, out, n, Di, Hi, Wi, precision, (hipStream_t)stream);
}

extern "C" size_t v3d_costreg_layer_split_workspace_bytes(int n, int cin, int Di, int Hi, int Wi) {
  if (n <= 0 || cin <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
  return v3d::align_up((size_t)n * cin * Di * Hi * Wi * sizeof(float), 256);
}

extern "C" int v3d_costreg_layer_split_f32(const v3d_costreg_weights* h, int layer, const float* in, const float* skip,
                                           int n, int Di, int Hi, int Wi, float* out, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  static const int cins[9] = {32, 8, 16, 16, 32, 32, 64, 64, 32};
  V3D_REQUIRE(h && in && out && workspace, V3D_ERR_BAD_ARG, "v3d_costreg_layer_split_f32: null argument");
  V3D_REQUIRE(layer >= 1 && layer <= 8, V3D_ERR_BAD_ARG, "v3d_costreg_layer_split_f32: layer %d (1..8: conv1..conv8)", layer);
  V3D_REQUIRE(n > 0 && Di > 0 && Hi > 0 && Wi > 0, V3D_ERR_BAD_SHAPE, "v3d_costreg_layer_split_f32: bad shape");
  V3D_REQUIRE(layer < 7 || skip, V3D_ERR_BAD_ARG, "v3d_costreg_layer_split_f32: conv7 / conv8 need the skip tensor");
  const int cin = cins[layer];
  V3D_REQUIRE(workspace_bytes >= v3d_costreg_layer_split_workspace_bytes(n, cin, Di, Hi, Wi), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_costreg_layer_split_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t plane = (size_t)Di * Hi * Wi, total = (size_t)n * (cin / 8) * plane;
  encode_split_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(in, (u32x4*)workspace, cin, plane, total);
  V3D_CHECK_LAUNCH("encode_split_kernel");
  const float* w = h->dev + (layer <= 6 ? h->cgbf_ofs[layer] : h->dgbf_ofs[layer - 7]);
  const float* bias = h->dev + h->bias_ofs[layer];
  switch (layer) {
    case 1: return launch_convg<CG<8, 16, 2, 14, kOutF32>>("costreg_conv1", workspace, w, bias, out, nullptr, n, Di, Hi, Wi, s);
    case 2: return launch_convg<CG<16, 16, 1, 14, kOutF32>>("costreg_conv2", workspace, w, bias, out, nullptr, n, Di, Hi, Wi, s);
    case 3: return launch_convg<CG<16, 32, 2, 14, kOutF32>>("costreg_conv3", workspace, w, bias, out, nullptr, n, Di, Hi, Wi, s);
    case 4: return launch_convg<CG<32, 32, 1, 14, kOutF32>>("costreg_conv4", workspace, w, bias, out, nullptr, n, Di, Hi, Wi, s);
    case 5: return launch_convg<CG<32, 64, 2, 8, kOutF32>>("costreg_conv5", workspace, w, bias, out, nullptr, n, Di, Hi, Wi, s);
    case 6: return launch_convg<CG<64, 64, 1, 8, kOutF32>>("costreg_conv6", workspace, w, bias, out, nullptr, n, Di, Hi, Wi, s);
    case 7: return launch_deconvg<DG<64, 32, 8, kOutF32>>("costreg_conv7", workspace, w, bias, skip, out, nullptr, n, Di, Hi, Wi, s);
    default: return launch_deconvg<DG<32, 16, 14, kOutF32>>("costreg_conv8", workspace, w, bias, skip, out, nullptr, n, Di, Hi, Wi, s);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// PropagationNet (mv3d/subnetworks/upsampling.py:14-36, stage 3 of eval-3dvnet.py:101-125):
//   x = cat(features, depth) -> 4 x [conv2d 3x3 p1 + BN + ReLU] (in -> 32 -> 32 -> 32 -> 9) -> softmax over the 9 logits
//   -> out = sum_k p_k * depth_pad[y + k / 3, x + k % 3]   (replicate padding).
// The four conv layers run on the FLAT variant of convg_bf16x2_kernel (split-bf16 matrix cores, the image stack as the
// z axis, split channel-last activations between the layers, BN folded, ReLU in the epilogue); the input is encoded and
// the softmax + 3x3 propagation applied by the two small kernels below.  Nothing runs on MIOpen / PyTorch.
// ---------------------------------------------------------------------------------------------------------------------
struct v3d_propagation_weights {
  int in_dim, cinp;
  float* dev;
  size_t w_ofs[4], b_ofs[4], total;
  size_t zw_ofs[4], zb_ofs[4], zw32_ofs[4];      // split-bf16 / exact-fp32 fragment images and padded biases of the row-marching kernel (propz.hip)
};

namespace {
// cat(features [B, Cf, HW], depth [B, HW]) -> split layout [CINP / 8][hi, lo][B * HW] (channels >= Cf + 1 are zero)
__global__ __launch_bounds__(256) void prop_encode_kernel(const float* __restrict__ feat, const float* __restrict__ depth,
                                                          u32x4* __restrict__ out, int Cf, int n_grp, size_t HW, size_t N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * n_grp) return;
  const size_t v = i % N;
  const int g = (int)(i / N);
  const size_t img = v / HW, px = v % HW;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = g * 8 + e;
    x[e] = c < Cf ? feat[(img * Cf + c) * HW + px] : c == Cf ? depth[v] : 0.f;
  }
  u32x2 ha, la, hb, lb;
  split4(x[0], x[1], x[2], x[3], ha, la);
  split4(x[4], x[5], x[6], x[7], hb, lb);
  out[((size_t)g * 2) * N + v] = (u32x4){ha.x, ha.y, hb.x, hb.y};
  out[((size_t)g * 2 + 1) * N + v] = (u32x4){la.x, la.y, lb.x, lb.y};
}

// softmax over the 9 (already ReLU'd) logits of a pixel (upsampling.py:27) and the weighted sum of its replicate-padded 3x3
// depth neighbourhood in unfold order (:29-36).  logits: [16][B * HW] (channels 9..15 are padding).
__global__ __launch_bounds__(256) void prop_finish_kernel(const float* __restrict__ logits, const float* __restrict__ depth,
                                                          float* __restrict__ out, int H, int W, size_t N) {
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= N) return;
  const size_t HW = (size_t)H * W, img = v / HW;
  const int y = (int)((v % HW) / W), x = (int)(v % W);
  float e[9], m = -3.4e38f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { e[k] = logits[(size_t)k * N + v]; m = fmaxf(m, e[k]); }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { e[k] = expf(e[k] - m); sum += e[k]; }
  const float* const d = depth + img * HW;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = min(max(y + k / 3 - 1, 0), H - 1), xx = min(max(x + k % 3 - 1, 0), W - 1);
    acc += (e[k] / sum) * d[(size_t)yy * W + xx];
  }
  out[v] = acc;
}
}  // namespace

extern "C" int v3d_propagation_pack(const float* const* conv_weight_host, const float* const* bn_weight_host,
                                    const float* const* bn_bias_host, const float* const* bn_mean_host,
                                    const float* const* bn_var_host, int in_dim, int h_dim, float bn_eps,
                                    v3d_propagation_weights** out_handle) {
  V3D_REQUIRE(conv_weight_host && bn_weight_host && bn_bias_host && bn_mean_host && bn_var_host && out_handle, V3D_ERR_BAD_ARG,
              "v3d_propagation_pack: null argument");
  V3D_REQUIRE(h_dim == 32, V3D_ERR_UNSUPPORTED, "v3d_propagation_pack: h_dim=%d unsupported (32, lightningmodel.py:41-43)", h_dim);
  const int cinp = (in_dim + 7) / 8 * 8;
  V3D_REQUIRE(in_dim >= 2 && (cinp == 8 || cinp == 24 || cinp == 40), V3D_ERR_UNSUPPORTED,
              "v3d_propagation_pack: in_dim=%d unsupported (guide channels + 1 <= 8, 17..24 or 33..40)", in_dim);
  auto* h = new v3d_propagation_weights();
  h->in_dim = in_dim; h->cinp = cinp; h->dev = nullptr;
  std::vector<float> host;
  auto reserve = [&](size_t n) { size_t o = host.size(); host.resize(o + (n + 63) / 64 * 64, 0.f); return o; };
  auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
  auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
  for (int l = 0; l < 4; ++l) {
    const int cin = l == 0 ? in_dim : 32, cout = l == 3 ? 9 : 32;
    const int cp = l == 0 ? cinp : 32, op = l == 3 ? 16 : 32, nch = cp / 8, ncg = op / 16;
    // [cout group][8-channel chunk][ky][hi, lo][lane 64][4 words]; rows = output channel, k = 8 * x tap + ci (x tap 3 = 0)
    h->w_ofs[l] = reserve((size_t)ncg * nch * 3 * 2 * 64 * 4);
    h->b_ofs[l] = reserve(op);
    unsigned* wb = reinterpret_cast<unsigned*>(host.data() + h->w_ofs[l]);
    for (int co = 0; co < cout; ++co) {
      const float sc = bn_weight_host[l][co] / sqrtf(bn_var_host[l][co] + bn_eps);
      host[h->b_ofs[l] + co] = bn_bias_host[l][co] - bn_mean_host[l][co] * sc;
    }
    for (int g = 0; g < ncg; ++g)
      for (int ch = 0; ch < nch; ++ch)
        for (int ky = 0; ky < 3; ++ky)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = g * 16 + (lane & 15), kx = lane >> 4;
            unsigned hi[8], lo[8];
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 8 + e;
              float v = 0.f;
              if (kx <= 2 && co < cout && ci < cin) {
                const float sc = bn_weight_host[l][co] / sqrtf(bn_var_host[l][co] + bn_eps);
                v = conv_weight_host[l][(((size_t)co * cin + ci) * 3 + ky) * 3 + kx] * sc;
              }
              hi[e] = rne(v);
              lo[e] = rne(v - up(hi[e]));
            }
            for (int part = 0; part < 2; ++part) {
              const unsigned* src = part ? lo : hi;
              unsigned* dst = wb + ((((size_t)g * nch + ch) * 3 + ky) * 2 + part) * 256 + lane * 4;
              for (int q = 0; q < 4; ++q) dst[q] = src[2 * q] | (src[2 * q + 1] << 16);
            }
          }
  }
  // the same layers for the row-marching kernel (propz.hip): K step = one tap x 32 input channels (layer 1: (tap, channel)
  // flattened), BN scale folded, biases padded to whole 16-row blocks
  for (int l = 0; l < 4; ++l) {
    const int cin = l == 0 ? in_dim : 32, cout = l == 3 ? 9 : 32, op = l == 3 ? 16 : 32;
    std::vector<float> wf((size_t)cout * cin * 9);
    for (int co = 0; co < cout; ++co) {
      const float sc = bn_weight_host[l][co] / sqrtf(bn_var_host[l][co] + bn_eps);
      for (int i = 0; i < cin * 9; ++i) wf[(size_t)co * cin * 9 + i] = conv_weight_host[l][(size_t)co * cin * 9 + i] * sc;
    }
    h->zw_ofs[l] = reserve(v3d::propz_image_words(l, cinp));
    h->zb_ofs[l] = reserve(op);
    h->zw32_ofs[l] = reserve(v3d::propz_image_words(l, cinp));
    v3d::propz_pack_layer(l, cinp, cin, cout, wf.data(), reinterpret_cast<unsigned*>(host.data() + h->zw_ofs[l]), false);
    v3d::propz_pack_layer(l, cinp, cin, cout, wf.data(), reinterpret_cast<unsigned*>(host.data() + h->zw32_ofs[l]), true);
    for (int co = 0; co < cout; ++co) host[h->zb_ofs[l] + co] = host[h->b_ofs[l] + co];
  }
  h->total = host.size();
  hipError_t e = hipMalloc((void**)&h->dev, h->total * sizeof(float));
  if (e != hipSuccess) { delete h; return v3d::fail(V3D_ERR_HIP, "hipMalloc(propagation weights): %s", hipGetErrorString(e)); }
  e = hipMemcpy(h->dev, host.data(), h->total * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return v3d::fail(V3D_ERR_HIP, "hipMemcpy(propagation weights): %s", hipGetErrorString(e)); }
  *out_handle = h;
  return V3D_OK;
}

extern "C" void v3d_propagation_free(v3d_propagation_weights* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  delete h;
}

extern "C" size_t v3d_propagation_workspace_bytes(const v3d_propagation_weights* h, int B, int H, int W) {
  if (!h || B <= 0 || H <= 0 || W <= 0) return 0;
  const size_t N = (size_t)B * H * W;
  // encoded input (cinp channels), two 32-channel activations (ping-pong), 16 logit channels: 4 bytes per value each
  return v3d::align_up((size_t)h->cinp * N * 4, 256) + 2 * v3d::align_up((size_t)32 * N * 4, 256) + v3d::align_up((size_t)16 * N * 4, 256);
}

extern "C" int v3d_propagation_up_f32(const v3d_propagation_weights* h, const float* features, const float* depth_lo, int B, int Cf,
                                      int H, int W, int h0, int w0, const int32_t* iy, const int32_t* ix, float* out, int precision,
                                      void* stream) {
  V3D_REQUIRE(h && features && depth_lo && out, V3D_ERR_BAD_ARG, "v3d_propagation_up_f32: null argument");
  V3D_REQUIRE(precision == V3D_PRECISION_SPLIT_BF16 || precision == V3D_PRECISION_FP32, V3D_ERR_BAD_ARG,
              "v3d_propagation_up_f32: unknown precision %d", precision);
  V3D_REQUIRE(B > 0 && H > 0 && W > 0 && h0 > 0 && w0 > 0 && Cf + 1 == h->in_dim, V3D_ERR_BAD_SHAPE,
              "v3d_propagation_up_f32: bad shape (B=%d, Cf=%d, H=%d, W=%d, depth %d x %d; packed for in_dim=%d)", B, Cf, H, W, h0, w0, h->in_dim);
  V3D_REQUIRE((iy && ix) || (!iy && !ix && h0 == H && w0 == W), V3D_ERR_BAD_ARG,
              "v3d_propagation_up_f32: both index tables, or none with a depth of the output size");
  const bool f32 = precision == V3D_PRECISION_FP32;
  const float* w[4]; const float* b[4];
  for (int l = 0; l < 4; ++l) { w[l] = h->dev + (f32 ? h->zw32_ofs[l] : h->zw_ofs[l]); b[l] = h->dev + h->zb_ofs[l]; }
  return v3d::launch_propz(h->cinp, f32, features, depth_lo, iy, ix, out, w, b, B, Cf, H, W, h0, w0, (hipStream_t)stream);
}

extern "C" int v3d_propagation_f32(const v3d_propagation_weights* h, const float* features, const float* depth, int B, int Cf,
                                   int H, int W, float* out, int precision, void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(h && features && depth && out && workspace, V3D_ERR_BAD_ARG, "v3d_propagation_f32: null argument");
  V3D_REQUIRE(precision == V3D_PRECISION_SPLIT_BF16 || precision == V3D_PRECISION_FP32, V3D_ERR_BAD_ARG,
              "v3d_propagation_f32: unknown precision %d", precision);
  V3D_REQUIRE(precision == V3D_PRECISION_SPLIT_BF16 || v3d::option(v3d::kOptPropFused) != 0, V3D_ERR_UNSUPPORTED,
              "v3d_propagation_f32: the per-layer kernels (option prop_fused = 0) have split-bf16 operands only");
  V3D_REQUIRE(B > 0 && H > 0 && W > 0 && Cf + 1 == h->in_dim, V3D_ERR_BAD_SHAPE,
              "v3d_propagation_f32: bad shape (B=%d, Cf=%d, H=%d, W=%d; packed for in_dim=%d)", B, Cf, H, W, h->in_dim);
  V3D_REQUIRE(workspace_bytes >= v3d_propagation_workspace_bytes(h, B, H, W), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_propagation_f32: workspace %zu < %zu", workspace_bytes, v3d_propagation_workspace_bytes(h, B, H, W));
  hipStream_t s = (hipStream_t)stream;
  if (v3d::option(v3d::kOptPropFused) != 0)      // one row-marching kernel, no workspace traffic (propz.hip)
    return v3d_propagation_up_f32(h, features, depth, B, Cf, H, W, H, W, nullptr, nullptr, out, precision, stream);
  const size_t N = (size_t)B * H * W;
  V3D_REQUIRE(N * 5 < ((size_t)1 << 31) * 4, V3D_ERR_BAD_SHAPE, "v3d_propagation_f32: batch too large (chunk the views)");
  char* base = (char*)workspace;
  void* enc = base;
  void* a = base + v3d::align_up((size_t)h->cinp * N * 4, 256);
  void* b = (char*)a + v3d::align_up((size_t)32 * N * 4, 256);
  float* logits = (float*)((char*)b + v3d::align_up((size_t)32 * N * 4, 256));
  {
    v3d::TimedScope ts("propagation_encode", s);
    const size_t total = N * (h->cinp / 8);
    prop_encode_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(features, depth, (u32x4*)enc, Cf, h->cinp / 8, (size_t)H * W, N);
  }
  V3D_CHECK_LAUNCH("prop_encode_kernel");
  auto Wt = [&](int l) { return h->dev + h->w_ofs[l]; };
  auto Bs = [&](int l) { return h->dev + h->b_ofs[l]; };
  int rc;
  if (h->cinp == 8) rc = launch_convg<CG<8, 32, 1, 14, kOutSplit, true>>("propagation_conv1", enc, Wt(0), Bs(0), nullptr, a, 1, B, H, W, s);
  else if (h->cinp == 24) rc = launch_convg<CG<24, 32, 1, 14, kOutSplit, true>>("propagation_conv1", enc, Wt(0), Bs(0), nullptr, a, 1, B, H, W, s);
  else rc = launch_convg<CG<40, 32, 1, 14, kOutSplit, true>>("propagation_conv1", enc, Wt(0), Bs(0), nullptr, a, 1, B, H, W, s);
  if (rc != V3D_OK) return rc;
  if ((rc = launch_convg<CG<32, 32, 1, 14, kOutSplit, true>>("propagation_conv2", a, Wt(1), Bs(1), nullptr, b, 1, B, H, W, s)) != V3D_OK) return rc;
  if ((rc = launch_convg<CG<32, 32, 1, 14, kOutSplit, true>>("propagation_conv3", b, Wt(2), Bs(2), nullptr, a, 1, B, H, W, s)) != V3D_OK) return rc;
  if ((rc = launch_convg<CG<32, 16, 1, 14, kOutF32, true>>("propagation_conv4", a, Wt(3), Bs(3), logits, nullptr, 1, B, H, W, s)) != V3D_OK) return rc;
  {
    v3d::TimedScope ts("propagation_softmax_sum", s);
    prop_finish_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(logits, depth, out, H, W, N);
  }
  V3D_CHECK_LAUNCH("prop_finish_kernel");
  return V3D_OK;
}

namespace {
struct WsPlan { size_t c0, c1, c2, c3, c4, c5, c6, u7, u8, u9, reg, enc, total; };
WsPlan plan_ws(int n, int D, int h, int w) {
  const size_t V0 = (size_t)D * h * w, V1 = V0 / 8, V2 = V1 / 8, V3 = V2 / 8;
  WsPlan p; size_t o = 0;
  auto take = [&](size_t nf) { size_t r = o; o += v3d::align_up(nf * sizeof(float), 256); return r; };
  p.c0 = take(n * 8 * V0); p.c1 = take(n * 16 * V1); p.c2 = take(n * 16 * V1);
  p.c3 = take(n * 32 * V2); p.c4 = take(n * 32 * V2); p.c5 = take(n * 64 * V3);
  p.c6 = take(n * 64 * V3); p.u7 = take(n * 32 * V2); p.u8 = take(n * 16 * V1);
  p.u9 = take(n * 8 * V0); p.reg = take(n * V0);
  p.enc = take(32 * V0);        // ONE view's variance volume in the split layout (fp32 entry point, see costreg_depth_impl)
  p.total = o;
  return p;
}
}  // namespace

extern "C" size_t v3d_costreg_workspace_bytes(const v3d_costreg_weights*, int n_ref, int D, int h, int w) {
  if (n_ref <= 0 || D <= 0 || h <= 0 || w <= 0) return 0;
  return plan_ws(n_ref, D, h, w).total;
}

namespace {
int launch_soft_argmin(const float* xreg, const float* depth_vals, float* depth, int n, int D, int H, int W, hipStream_t s) {
  const size_t npix = (size_t)n * H * W;
  {
    v3d::TimedScope ts("soft_argmin", s);
    soft_argmin_kernel<<<(unsigned)((npix + 255) / 256), 256, 0, s>>>(xreg, depth_vals, depth, n, D, H * W);
  }
  V3D_CHECK_LAUNCH("soft_argmin_kernel");
  return V3D_OK;
}

// conv9 + skip + prob: the tile kernel (split-bf16 or exact-fp32 operands)
int launch_conv9_prob(bool f32, const v3d_costreg_weights* h, const float* u8, const float* c0, float* xreg, int n, int D, int H, int W,
                      hipStream_t s) {
  C9Params q;
  q.u8 = u8; q.c0 = c0; q.wbf = h->dev + (f32 ? h->c9f32_ofs : h->c9bf_ofs); q.bias9 = h->dev + h->bias_ofs[9];
  q.wprob = h->dev + h->prob_w2_ofs; q.bprob = h->dev + h->prob_b_ofs; q.out = xreg;
  q.n = n; q.D = D; q.H = H; q.W = W;
  q.ntz = (D + C9::TD - 1) / C9::TD; q.nty = (H + C9::TH - 1) / C9::TH; q.ntx = (W + C9::TW - 1) / C9::TW;
  q.zy_order = tile_order(q.ntx, q.nty);
  const long long blocks = (long long)n * q.ntz * q.nty * q.ntx;
  V3D_REQUIRE(blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv9+prob: grid too large");
  // 32-bit byte offsets inside one view's tensors (conv0 skip: 2 x 16 bytes per voxel / 8 fp32 planes)
  V3D_REQUIRE((long long)D * H * W < (1ll << 26) && D < (1 << 12) && H < (1 << 12) && W < (1 << 12), V3D_ERR_BAD_SHAPE,
              "conv9+prob: volume %d x %d x %d too large for 32-bit offsets", D, H, W);
  q.m_tx = v3d::magic_u32((unsigned long long)blocks, (unsigned)q.ntx);
  q.m_t1 = v3d::magic_u32((unsigned long long)blocks / q.ntx + 1, (unsigned)(q.zy_order ? q.ntz : q.nty));
  q.m_t2 = v3d::magic_u32((unsigned long long)blocks / q.ntx / (q.zy_order ? q.ntz : q.nty) + 1, (unsigned)(q.zy_order ? q.nty : q.ntz));
  {
    v3d::TimedScope ts(f32 ? "costreg_conv9_prob_f32" : "costreg_conv9_prob", s);
    // (round 6: a persistent variant -- two workgroups per CU walking their tiles, the weight fragments resident in their own
    // 18 KB of LDS, the next tile's input slots requested behind the staging barrier / the matrix phase / the u9 assembly --
    // measured 0.49-0.52 ms against the 0.43-0.45 of one workgroup per tile on the same box, wherever the prefetch sat: DESIGN.md 8.3)
    if (f32) conv9_prob_kernel<true><<<(unsigned)blocks, 512, 0, s>>>(q);
    else conv9_prob_kernel<false><<<(unsigned)blocks, 512, 0, s>>>(q);
  }
  V3D_CHECK_LAUNCH("conv9_prob_kernel");
  return V3D_OK;
}

// Side streams of the regulariser's tail (one set per device, created on first use, never destroyed: the library's only
// device-side resources besides weight handles).  Non-blocking streams; ordering against the caller's stream is by events.
constexpr int kTailStreamsMax = 8;
struct TailStreams { hipStream_t st[kTailStreamsMax]; hipEvent_t fork, done[kTailStreamsMax]; int n; };
TailStreams* tail_streams(int want) {
  static TailStreams pool[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  TailStreams& t = pool[dev];
  if (t.n == 0 && hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
  while (t.n < want) {
    if (hipStreamCreateWithFlags(&t.st[t.n], hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&t.done[t.n], hipEventDisableTiming) != hipSuccess) return nullptr;
    ++t.n;
  }
  return &t;
}

// The split-bf16 regulariser behind conv0: conv1 + conv2 (step 1), conv3 .. conv8 (steps 3 .. 8), conv9 + skip + prob (9),
// soft-argmin (10), for views [v0, v0 + nv) on stream st.  Every tensor of the chain is per-view contiguous, so a sub-batch
// is a pointer offset.
struct TailCtx {
  const v3d_costreg_weights* h; WsPlan ws; char* base; const float* depth_vals; float* depth; float* xreg; int n, D, H, W;
};
int run_tail_steps(const TailCtx& c, int first, int last, int v0, int nv, hipStream_t st) {
  const v3d_costreg_weights* h = c.h;
  const int D = c.D, H = c.H, W = c.W;
  const size_t V0 = (size_t)D * H * W, V1 = V0 / 8, V2 = V1 / 8, V3 = V2 / 8;
  auto F = [&](size_t o, size_t per_view) { return (float*)(c.base + o) + (size_t)v0 * per_view; };
  // conv1..conv6 hand their activations on in the split layout; the transposed convolutions read their skips (conv2, conv4)
  // from the same split copies (DG::SKIP_SPLIT).  The split copies live in the u9 slot of the workspace, which the fused
  // conv9+prob kernel does not need.
  float* const c2s_all = (float*)(c.base + c.ws.u9);
  float* const c4s_all = c2s_all + (size_t)c.n * 16 * V1;
  float* const c2s = c2s_all + (size_t)v0 * 16 * V1;
  float* const c4s = c4s_all + (size_t)v0 * 32 * V2;
  auto W_ = [&](int l) { return h->dev + h->cgbf_ofs[l]; };
  auto B_ = [&](int l) { return h->dev + h->bias_ofs[l]; };
  int rc;
#ifdef V3D_PHASE_TIMING
  const int stop_after = v3d::option(v3d::kOptStopAfter);   // isolate one kernel's counters
#else
  const int stop_after = 99;
#endif
  for (int step = first; step <= last; ++step) {
    if (step > stop_after) return V3D_OK;
    switch (step) {
      case 1:
        // conv1 + conv2: the fused depth march of conv12z.hip (conv1's output never leaves LDS: 0.27 ms per 64 cfg2 views against
        // 0.176 + 0.146 for the two tile kernels, which remain behind the developer option c12_march = 0 and the per-layer entry points)
        if (v3d::option(v3d::kOptC12March) != 0) {
          if ((rc = v3d::launch_conv12z(F(c.ws.c0, 8 * V0), W_(1), W_(2), B_(1), B_(2), c2s, nv, D, H, W, st)) != V3D_OK) return rc;
        } else {
          if ((rc = launch_convg<CG<8, 16, 2, 14, kOutSplit>>("costreg_conv1", F(c.ws.c0, 8 * V0), W_(1), B_(1), nullptr, F(c.ws.c1, 16 * V1),
                                                              nv, D, H, W, st)) != V3D_OK) return rc;
          if ((rc = launch_convg<CG<16, 16, 1, 14, kOutSplit>>("costreg_conv2", F(c.ws.c1, 16 * V1), W_(2), B_(2), nullptr, c2s, nv, D / 2,
                                                               H / 2, W / 2, st)) != V3D_OK) return rc;
        }
        break;
      case 3:
        if ((rc = launch_convg<CG<16, 32, 2, 14, kOutSplit>>("costreg_conv3", c2s, W_(3), B_(3), nullptr, F(c.ws.c3, 32 * V2), nv, D / 2,
                                                             H / 2, W / 2, st)) != V3D_OK) return rc;
        break;
      case 4:
        if ((rc = launch_convg<CG<32, 32, 1, 14, kOutSplit>>("costreg_conv4", F(c.ws.c3, 32 * V2), W_(4), B_(4), nullptr, c4s, nv, D / 4,
                                                             H / 4, W / 4, st)) != V3D_OK) return rc;
        break;
      case 5:
        if ((rc = launch_convg<CG<32, 64, 2, 8, kOutSplit>>("costreg_conv5", c4s, W_(5), B_(5), nullptr, F(c.ws.c5, 64 * V3), nv, D / 4,
                                                            H / 4, W / 4, st)) != V3D_OK) return rc;
        break;
      case 6:
        if ((rc = launch_convg<CG<64, 64, 1, 8, kOutSplit>>("costreg_conv6", F(c.ws.c5, 64 * V3), W_(6), B_(6), nullptr, F(c.ws.c6, 64 * V3),
                                                            nv, D / 8, H / 8, W / 8, st)) != V3D_OK) return rc;
        break;
      case 7:      // conv4 + conv7(x) (mvsnet.py:159)
        if ((rc = launch_deconvg<DG<64, 32, 8, kOutSplit, true>>("costreg_conv7", F(c.ws.c6, 64 * V3), h->dev + h->dgbf_ofs[0], B_(7), c4s,
                                                                 nullptr, F(c.ws.u7, 32 * V2), nv, D / 8, H / 8, W / 8, st)) != V3D_OK) return rc;
        break;
      case 8:      // conv2 + conv8(x) (:160)
        if ((rc = launch_deconvg<DG<32, 16, 14, kOutSplit, true>>("costreg_conv8", F(c.ws.u7, 32 * V2), h->dev + h->dgbf_ofs[1], B_(8), c2s,
                                                                  nullptr, F(c.ws.u8, 16 * V1), nv, D / 4, H / 4, W / 4, st)) != V3D_OK) return rc;
        break;
      case 9:
        // conv9 + skip + prob: the tile kernel.  Developer A/B (v3d_set_option "c9_kernel" = 1): the depth-march experiment of
        // round 4 (conv9z.hip: correct, but 0.51 against 0.46 ms per 64 views, see its header; -DV3D_EXPERIMENTS builds only)
        if (v3d::option(v3d::kOptC9Kernel) == 1) {
          if ((rc = v3d::launch_conv9z(F(c.ws.u8, 16 * V1), F(c.ws.c0, 8 * V0), h->dev + h->c9bf_ofs, h->dev + h->bias_ofs[9],
                                       h->dev + h->prob_w2_ofs, h->dev + h->prob_b_ofs, c.xreg + (size_t)v0 * V0, nv, D, H, W, st)) != V3D_OK)
            return rc;
        } else if ((rc = launch_conv9_prob(false, h, F(c.ws.u8, 16 * V1), F(c.ws.c0, 8 * V0), c.xreg + (size_t)v0 * V0, nv, D, H, W, st)) != V3D_OK) {
          return rc;
        }
        break;
      case 10:
        if ((rc = launch_soft_argmin(c.xreg + (size_t)v0 * V0, c.depth_vals, c.depth + (size_t)v0 * H * W, nv, D, H, W, st)) != V3D_OK) return rc;
        break;
      default: break;      // step 2: conv2 is part of step 1
    }
  }
  return V3D_OK;
}

// Views are independent through the whole regulariser, and the layers behind conv0 are latency-bound rather than
// throughput-bound (grids of a few hundred short workgroups, 0.04-0.1 ms each: DESIGN.md 8.3).  Steps [tail_from, tail_to]
// therefore run as `tail_streams` sub-batches of views on the library's side streams -- the kernels of different sub-batches
// fill each other's ramp-up, drain and barrier gaps -- forked from and joined back into the caller's stream by events
// (capturable in a HIP graph: a fork / join inside one capture).  Same kernels on the same data: bit-identical results.
int run_split_tail(const v3d_costreg_weights* h, const WsPlan& ws, char* base, const float* depth_vals, int n, int D, int H, int W,
                   float* depth, float* xreg, hipStream_t s) {
  TailCtx c{h, ws, base, depth_vals, depth, xreg, n, D, H, W};
  int S = v3d::option(v3d::kOptTailStreams);
  int first = v3d::option(v3d::kOptTailFrom), last = v3d::option(v3d::kOptTailTo);
  V3D_REQUIRE(S >= 1 && S <= kTailStreamsMax && first >= 1 && first <= 10 && last >= first && last <= 10, V3D_ERR_BAD_ARG,
              "options tail_streams (1..%d) / tail_from / tail_to (1..10) out of range", kTailStreamsMax);
  if (S > n) S = n;
#ifdef V3D_PHASE_TIMING
  S = 1;
#endif
  int rc;
  if (S == 1) return run_tail_steps(c, 1, 10, 0, n, s);
  TailStreams* t = tail_streams(S);
  V3D_REQUIRE(t, V3D_ERR_HIP, "the regulariser's side streams could not be created");
  if ((rc = run_tail_steps(c, 1, first - 1, 0, n, s)) != V3D_OK) return rc;
  V3D_CHECK_HIP(hipEventRecord(t->fork, s));
  for (int k = 0; k < S; ++k) {
    const int v0 = (int)((long long)n * k / S), v1 = (int)((long long)n * (k + 1) / S);
    V3D_CHECK_HIP(hipStreamWaitEvent(t->st[k], t->fork, 0));
    if ((rc = run_tail_steps(c, first, last, v0, v1 - v0, t->st[k])) != V3D_OK) return rc;
    V3D_CHECK_HIP(hipEventRecord(t->done[k], t->st[k]));
  }
  for (int k = 0; k < S; ++k) V3D_CHECK_HIP(hipStreamWaitEvent(s, t->done[k], 0));
  return run_tail_steps(c, last + 1, 10, 0, n, s);
}
}  // namespace

// in_layout: 0 = reference fp32 [n, C, D, h, w], 1 = split-bf16 hand-off, 2 = fp32 channel-last (v3d_psv_variance_cl8)
static int costreg_depth_impl(int in_layout, const v3d_costreg_weights* h, const float* var,
                              const float* depth_vals, int n, int D, int H, int W,
                              float* depth, float* reg, int precision, void* workspace,
                              size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(precision == V3D_PRECISION_SPLIT_BF16 || precision == V3D_PRECISION_FP32, V3D_ERR_BAD_ARG,
              "v3d_costreg_depth_f32: unknown precision %d", precision);
  V3D_REQUIRE(h && var && depth_vals && depth && workspace, V3D_ERR_BAD_ARG,
              "v3d_costreg_depth_f32: null argument");
  V3D_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && D % 8 == 0 && H % 8 == 0 && W % 8 == 0,
              V3D_ERR_BAD_SHAPE, "v3d_costreg_depth_f32: D,h,w must be positive multiples of 8 (got %d,%d,%d)", D, H, W);
  const WsPlan ws = plan_ws(n, D, H, W);
  V3D_REQUIRE(workspace_bytes >= ws.total, V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_costreg_depth_f32: workspace %zu < %zu", workspace_bytes, ws.total);
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  auto F = [&](size_t o) { return (float*)(base + o); };
  float* xreg = reg ? reg : F(ws.reg);
  int rc;
#define RUN(layer, in, skip, out, d, hh, ww) \
  if ((rc = run_layer(h, layer, in, skip, out, n, d, hh, ww, V3D_PRECISION_FP32, s)) != V3D_OK) return rc;
  // precision == V3D_PRECISION_FP32 (fp32 volume only): every layer on the exact-fp32 per-layer kernels with fp32
  // [n, C, D, H, W] tensors in between.  V3D_PRECISION_SPLIT_BF16: every layer on split-bf16 matrix cores, activations
  // in the split channel-last hand-off format (same bytes as fp32).
  const bool generic = precision == V3D_PRECISION_FP32;
  const bool split_in = in_layout == 1, cl8_in = in_layout == 2;
  V3D_REQUIRE(!generic || !split_in, V3D_ERR_UNSUPPORTED, "V3D_PRECISION_FP32 needs an fp32 variance volume");
  V3D_REQUIRE(h->in_channels == 32 || in_layout == 0, V3D_ERR_UNSUPPORTED,
              "CostRegNet(16, 8) takes the variance volume in the reference layout (the hand-off formats are defined for 32 channels)");
  V3D_REQUIRE(generic || !cl8_in, V3D_ERR_UNSUPPORTED, "the fp32 channel-last volume is the input of V3D_PRECISION_FP32");
  if (generic) {
    if (cl8_in) {      // conv0 as a depth march on exact-fp32 matrix instructions (conv0z.hip)
      if ((rc = v3d::launch_conv0z(true, var, h->dev + h->c0f32_ofs, h->dev + h->bias_ofs[0], F(ws.c0), n, D, H, W, s)) != V3D_OK)
        return rc;
    } else {
      RUN(0, var, nullptr, F(ws.c0), D, H, W);
    }
    RUN(1, F(ws.c0), nullptr, F(ws.c1), D, H, W);
    RUN(2, F(ws.c1), nullptr, F(ws.c2), D / 2, H / 2, W / 2);
  } else {
    // conv0: the depth-march kernel (conv0z.hip) reads the split hand-off format.  The fp32 entry point (CostRegNet.forward
    // on a reference-layout tensor, return_intermediates) encodes one view at a time into the workspace's `enc` slot and runs
    // the same kernel on it: both entry points give the same bits.
    if (split_in) {
      if ((rc = v3d::launch_conv0z(false, var, h->dev + h->c0bf_ofs, h->dev + h->bias_ofs[0], F(ws.c0), n, D, H, W, s)) != V3D_OK) return rc;
    } else {
      // (CostRegNet(16, 8): channel groups 2, 3 of the encoded view are zeros, once -- their conv0 weights are zero as well)
      const int cin = h->in_channels, ngrp = cin / 8;
      const size_t V0 = (size_t)D * H * W, total = (size_t)ngrp * V0;
      if (ngrp < 4) V3D_CHECK_HIP(hipMemsetAsync((char*)F(ws.enc) + (size_t)ngrp * 2 * V0 * 16, 0, (size_t)(4 - ngrp) * 2 * V0 * 16, s));
      for (int i = 0; i < n; ++i) {
        encode_split_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(var + (size_t)i * cin * V0, (u32x4*)F(ws.enc), cin, V0, total);
        V3D_CHECK_LAUNCH("encode_split_kernel");
        if ((rc = v3d::launch_conv0z(false, F(ws.enc), h->dev + h->c0bf_ofs, h->dev + h->bias_ofs[0], F(ws.c0) + (size_t)i * 8 * V0, 1, D, H,
                                     W, s)) != V3D_OK) return rc;
      }
    }
    // conv1 .. conv9 + prob, soft-argmin: see run_split_tail() below (sub-batches of views on concurrent streams)
    return run_split_tail(h, ws, base, depth_vals, n, D, H, W, depth, xreg, s);
  }
  RUN(3, F(ws.c2), nullptr, F(ws.c3), D / 2, H / 2, W / 2);
  RUN(4, F(ws.c3), nullptr, F(ws.c4), D / 4, H / 4, W / 4);
  RUN(5, F(ws.c4), nullptr, F(ws.c5), D / 4, H / 4, W / 4);
  RUN(6, F(ws.c5), nullptr, F(ws.c6), D / 8, H / 8, W / 8);
  RUN(7, F(ws.c6), F(ws.c4), F(ws.u7), D / 8, H / 8, W / 8);    // conv4 + conv7(x)  (mvsnet.py:159)
  RUN(8, F(ws.u7), F(ws.c2), F(ws.u8), D / 4, H / 4, W / 4);    // conv2 + conv8(x)  (:160)
  // conv9 + skip + prob on exact-fp32 operands: the tile kernel; developer A/B (v3d_set_option "c9_kernel" = 2): the per-layer
  // conv9 kernel + prob_conv_kernel (rounds 1-3)
  if (v3d::option(v3d::kOptC9Kernel) != 2) {
    if ((rc = launch_conv9_prob(true, h, F(ws.u8), F(ws.c0), xreg, n, D, H, W, s)) != V3D_OK) return rc;
  } else {
    RUN(9, F(ws.u8), F(ws.c0), F(ws.u9), D / 2, H / 2, W / 2);    // conv0 + conv9(x)  (:161)
    {
      v3d::TimedScope ts("costreg_prob", s);
      const int ntz = (D + PT_D - 1) / PT_D, nty = (H + PT_H - 1) / PT_H, ntx = (W + PT_W - 1) / PT_W;
      prob_conv_kernel<8><<<(unsigned)((size_t)n * ntz * nty * ntx), 256, 0, s>>>(      // W % 8 == 0 (checked above)
          F(ws.u9), h->dev + h->prob_w_ofs, h->dev + h->prob_b_ofs, xreg, n, D, H, W, ntz, nty, ntx);
    }
    V3D_CHECK_LAUNCH("prob_conv_kernel");
  }
#undef RUN
  return launch_soft_argmin(xreg, depth_vals, depth, n, D, H, W, s);
}

extern "C" int v3d_costreg_depth_f32(const v3d_costreg_weights* h, const float* var, const float* depth_vals, int n,
                                     int D, int H, int W, float* depth, float* reg, int precision, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return costreg_depth_impl(0, h, var, depth_vals, n, D, H, W, depth, reg, precision, workspace, workspace_bytes,
                            stream);
}

extern "C" int v3d_costreg_depth_cl8(const v3d_costreg_weights* h, const void* var_cl8, const float* depth_vals,
                                     int n, int D, int H, int W, float* depth, float* reg, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return costreg_depth_impl(2, h, (const float*)var_cl8, depth_vals, n, D, H, W, depth, reg, V3D_PRECISION_FP32, workspace,
                            workspace_bytes, stream);
}

extern "C" int v3d_costreg_depth_split(const v3d_costreg_weights* h, const void* var_split, const float* depth_vals,
                                       int n, int D, int H, int W, float* depth, float* reg, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return costreg_depth_impl(1, h, (const float*)var_split, depth_vals, n, D, H, W, depth, reg,
                            V3D_PRECISION_SPLIT_BF16, workspace, workspace_bytes, stream);
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)8 * kPhaseSlots);
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_phase), h.size() * sizeof(unsigned long long)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < kPhaseSlots; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif
