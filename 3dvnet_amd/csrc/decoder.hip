// Row C2b tail + C3 of SURVEY.md §8a: the hypothesis decoder's last Conv1d(h_dim -> 1, k3, pad 1, bias)
// along the hypothesis axis, softmax over the hypotheses (refinement.py:24,43) and, optionally, the
// expected depth offset sum_i p_i * vals_i (lightningmodel.py:238-241).  One wave per point.
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gemm_weights.h"
#include "sparse_hash.h"

namespace {

constexpr int kMaxHyp = 16;

__global__ __launch_bounds__(256) void decoder_head_kernel(const float* __restrict__ act, int n_pts,
                                                           int n_hyp, int C, const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ vals,
                                                           float* __restrict__ preds,
                                                           float* __restrict__ expect) {
  const int lane = threadIdx.x & 63;
  const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= n_pts) return;
  float score[kMaxHyp];
#pragma unroll
  for (int h = 0; h < kMaxHyp; ++h) score[h] = 0.f;
  const float* a = act + (size_t)pt * n_hyp * C;
  for (int c = lane; c < C; c += 64) {
    const float w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];   // weight [1, C, 3]
#pragma unroll
    for (int h = 0; h < kMaxHyp; ++h) {
      if (h < n_hyp) {
        const float x = a[(size_t)h * C + c];
        // out[h'] = sum_t in[h' + t - 1] w[t]  =>  in[h] feeds out[h+1] (t=0), out[h] (t=1), out[h-1] (t=2)
        if (h + 1 < n_hyp) score[h + 1] += x * w0;
        score[h] += x * w1;
        if (h > 0) score[h - 1] += x * w2;
      }
    }
  }
#pragma unroll
  for (int h = 0; h < kMaxHyp; ++h) {
    float v = score[h];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    score[h] = v + bias[0];
  }
  if (lane == 0) {
    float m = -INFINITY;
    for (int h = 0; h < n_hyp; ++h) m = fmaxf(m, score[h]);
    float s = 0.f;
    for (int h = 0; h < n_hyp; ++h) s += expf(score[h] - m);
    float e = 0.f;
    for (int h = 0; h < n_hyp; ++h) {
      const float pr = expf(score[h] - m) / s;
      preds[(size_t)pt * n_hyp + h] = pr;
      if (vals) e += vals[h] * pr;
    }
    if (expect) expect[pt] = e;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused hypothesis decoder (rows C2a + C2b + C3): sparse trilinear interpolation of the three U-Net
// levels (refinement.py:28-41) -> Conv1d+BN+ReLU x3 along the hypothesis axis (:16-23) -> Conv1d(128 -> 1) + softmax (:24,43)
// -> expected offset (lightningmodel.py:237-241).  The [Nq, 352, 7] feature tensor (30.9 MB per reference view and sweep) and
// the three [Nq*7, 128] activations never reach HBM.
//
// Round 5 decomposition (the round-3/4 kernel kept activations in LDS, every wave fetched its own weight fragments through the
// CU's vector-memory path -- 0.93 of the 1.6 MB a 64-column tile moved through it -- and spilled 47 registers):
//
//   * COLUMN-OWNER WAVES.  A wave owns 32 MFMA columns = 4 query points x 8 hypothesis slots (n_hyp <= 8; slot h >= n_hyp is
//     padding) and ALL 128 output channels: four v_mfma_f32_32x32x16_bf16 row blocks, 64 accumulator registers.  The C/D layout
//     of that instruction gives lane (g = lane >> 5, n = lane & 31) the channels 32 mb + 8 j + 4 g + r of column n -- exactly
//     the 8 k-values per 16-wide K step a B fragment wants once the NEXT layer's weights are packed in that channel order
//     (v3d_gemm_pack, dec_ofs).  So bias + ReLU + hi/lo split turn a layer's accumulators into the next layer's B fragments
//     in registers: no activation ever touches LDS, no barrier belongs to the data path.
//   * The conv taps h - 1 / h + 1 are lane shifts of the B fragment inside a 16-lane row (DPP row_shr / row_shl, zero fill),
//     masked where a tap leaves the hypothesis group.
//   * WEIGHTS THROUGH AN LDS RING.  The 8 waves of the workgroup (one per CU, 256 columns = 32 points per tile) walk the
//     (layer, 16-channel K step) sequence together; the step's 24 KB slab ([3 taps][hi, lo][4 row blocks] A fragments) is
//     brought into a 3-slot LDS ring by LDS-DMA two steps ahead (3 x 1 KB per wave), one s_barrier per step: the vector-memory
//     path carries each weight byte once per 256 columns instead of once per 64, and A fragments are ds_read_b128 (LDS has
//     four times the bandwidth of that path).
//   * Layer 1's B fragments are PRODUCED by the lane that consumes them: lane (g, n) gathers the 8 corner rows of column n
//     for its 8 channels of the step (2 x 16 B per corner; the two g halves of a column read 32 contiguous bytes), blends,
//     splits.  The gathers of step s + 1 are in flight during the matrix instructions of step s.
//   * The hash probes moved into a pre-kernel (decoder_corner_kernel: one thread per (query row, corner), three levels) that
//     leaves (row, weight) pairs in a caller-provided table; a data-dependent probe loop inside barrier-coupled matrix waves
//     stalled all of them.  The tile's entries are parked in a wave-private LDS table during layer 2 of the previous tile.
//
// Matrix arithmetic as everywhere on this path: split-bf16 operands (hi*hi + hi*lo + lo*hi), fp32 accumulation.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kDWaves = 8, kDThreads = 64 * kDWaves;
constexpr int kDPtsWave = 4, kDPtsTile = kDPtsWave * kDWaves;     // query points per wave / per tile (8 column slots each)
constexpr int kDH = 128;                                           // hidden width of the decoder
constexpr int kDSlabFrags = 24;                                    // [3 taps][hi, lo][4 row blocks] fragments of 1 KB
constexpr int kDSlab = kDSlabFrags * 1024;                         // bytes per 16-channel K step
constexpr int kDRing = 3;
constexpr int kDCtabWave = 4 * 8 * 32 * 8;                         // bytes: [3 levels + point features][corner][column] (row, weight)
constexpr int kDStageWave = 2 * 64 * 16;                           // bytes: [hi, lo][g][column] one B fragment slot each
constexpr int kDLdsRing = kDRing * kDSlab;
constexpr int kDLdsCtab = kDWaves * kDCtabWave;
constexpr int kDLdsStage = kDWaves * kDStageWave;
constexpr int kDLdsConst = (3 * kDH + 3 * kDH + 16) * 4;           // permuted biases, permuted head weights, head bias, offset values
constexpr int kDLdsBytes = kDLdsRing + kDLdsCtab + kDLdsStage + kDLdsConst;     // 158 784 B: one workgroup per CU
static_assert(kDLdsBytes <= 160 * 1024, "LDS of one CU");
static_assert(kDSlabFrags % kDWaves == 0, "every wave issues the same number of LDS-DMA pieces per step");
constexpr int kDPieces = kDSlabFrags / kDWaves;                    // 3

struct CornerParams {
  v3dhash::HashTable table[3];
  const float* min_pts[3];     // [n_batch, 3]
  float res[3];                // x.res of the level (= tensor_stride * voxel size)
  float tsf[3], inv_ts[3];     // tensor stride and its reciprocal (exact when the stride is a power of two)
  const float* pts;            // [n_q, 3]
  const long long* pts_batch;  // [n_pts]
  int n_hyp;
  unsigned m_hyp;              // v3d::magic_u32 of n_hyp
  int n_q;
  u32x2* out;                  // [n_q][3 levels][8 corners] (feature row, weight bits); absent corner = (0, 0.f)
};

#ifdef V3D_CORNER_STATS
// developer build only (scripts/micro/corner_stats.py): [level][0 lookups, 1 probes, 2 longest chain, 3 present]
__device__ unsigned long long g_corner_stats[12];
__device__ __forceinline__ int corner_find_counted(const v3dhash::HashTable& t, unsigned long long key, int l) {
  unsigned slot = v3dhash::hash_u64(key) & t.mask;
  int row = -1;
  unsigned probe = 0;
  for (; probe <= t.mask; ++probe) {
    const unsigned long long k = t.entries[slot].key;
    if (k == key) { row = t.entries[slot].val; break; }
    if (k == v3dhash::kEmpty) break;
    slot = (slot + 1) & t.mask;
  }
  atomicAdd(&g_corner_stats[l * 4 + 0], 1ull);
  atomicAdd(&g_corner_stats[l * 4 + 1], (unsigned long long)(probe + 1));
  atomicMax(&g_corner_stats[l * 4 + 2], (unsigned long long)(probe + 1));
  if (row >= 0) atomicAdd(&g_corner_stats[l * 4 + 3], 1ull);
  return row;
}
#endif

// Rows C2a's index half: the 8 lattice corners of every hypothesis point on the three levels (refinement.py:33-39 feeding
// MinkowskiInterpolation): query coordinate ((p - min) / x.res) * x.stride in base-voxel units, corner floor(q / ts) ts + {0, ts}^3,
// weight prod (1 - |q - c| / ts), absent corners contribute nothing (no renormalisation).  One thread per (row, corner), the three
// levels in turn.  POW2: the tensor strides are powers of two (the U-Net's 1, 2, 4): x / ts == x * (1 / ts) bit for bit, six of the
// nine IEEE divisions per level become multiplications.  What bounds the kernel is the number of L2 requests, not the instructions:
// 34 M random 8-byte key reads + 16 M value reads per 64-view sweep in 0.23 ms are ~0.2 T requests/s, the rate of the L2 channels
// (one thread per (row, level, corner) -- a third of the instructions per thread, three times the threads re-reading the point --
// measured 0.28 ms against 0.23).
template <bool POW2>
__global__ __launch_bounds__(256) void decoder_corner_kernel(CornerParams cp) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;      // (host: 8 n_q < 2^31)
  const unsigned q = gid >> 3;
  const int corner = (int)(gid & 7u);
  if (q >= (unsigned)cp.n_q) return;
  const int b = (int)cp.pts_batch[v3d::udiv_magic(q, (unsigned)cp.n_hyp, cp.m_hyp)];
  const float px = cp.pts[(size_t)q * 3 + 0], py = cp.pts[(size_t)q * 3 + 1], pz = cp.pts[(size_t)q * 3 + 2];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float ts = cp.tsf[l], res = cp.res[l], its = cp.inv_ts[l];
    const float* mn = cp.min_pts[l] + b * 3;
    const float qx = ((px - mn[0]) / res) * ts, qy = ((py - mn[1]) / res) * ts, qz = ((pz - mn[2]) / res) * ts;
    auto over_ts = [&](float x) __attribute__((always_inline)) { return POW2 ? x * its : x / ts; };
    const float c0 = floorf(over_ts(qx)) * ts + ((corner & 1) ? ts : 0.f);
    const float c1 = floorf(over_ts(qy)) * ts + ((corner & 2) ? ts : 0.f);
    const float c2 = floorf(over_ts(qz)) * ts + ((corner & 4) ? ts : 0.f);
    float w = 1.f;
    w *= 1.f - over_ts(fabsf(qx - c0));
    w *= 1.f - over_ts(fabsf(qy - c1));
    w *= 1.f - over_ts(fabsf(qz - c2));
    int row = -1;
    // (a coordinate outside the key range cannot be present and is never looked up)
    if (c0 >= -v3dhash::kGuard && c1 >= -v3dhash::kGuard && c2 >= -v3dhash::kGuard && c0 <= 60000.f && c1 <= 60000.f && c2 <= 60000.f) {
#ifdef V3D_CORNER_STATS
      row = corner_find_counted(cp.table[l], v3dhash::pack_key(b, (int)c0, (int)c1, (int)c2), l);
#else
      row = v3dhash::hash_find(cp.table[l], v3dhash::pack_key(b, (int)c0, (int)c1, (int)c2));
#endif
    }
    // an absent corner reads feature row 0 with weight 0: the gathers of the fused kernel are unconditional
    cp.out[((size_t)q * 3 + l) * 8 + corner] = (u32x2){row < 0 ? 0u : (unsigned)row, row < 0 ? 0u : __float_as_uint(w)};
  }
}

struct FusedParams {
  const float* feats[3];      // level features [N_l, C_l], in feature-row order: channels [0, C0) = finest level, then the coarser two
  int C[3];
  const u32x2* ctab;          // decoder_corner_kernel's table
  const float* pts_feat;      // [n_pts, n_hyp, c_feat] or null
  int c_feat, n_pts, n_hyp;
  const void* w[3];           // slab images (v3d_gemm_weights::dec_ofs) of the three Conv1d layers
  int nstep1;                 // 16-channel K steps of layer 1 (the 128 -> 128 layers have 8)
  const float* bias[3];       // folded BatchNorm biases [128]
  const float* head_w;        // [1, 128, 3]
  const float* head_b;
  const float* vals;          // [n_hyp] offset values or null
  float* preds;               // [n_pts, n_hyp]
  float* expect;              // [n_pts] or null
  float* depth_io;            // [n_pts] or null: depth_io[pt] += expected offset (the caller's `depth += offset`, eval-3dvnet.py:99)
};

__device__ __forceinline__ unsigned fused_pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
// x = hi + lo (hi = RNE_bf16(x), lo = RNE_bf16(x - hi)) for 8 values -> four words of hi, four of lo (one B fragment each)
__device__ __forceinline__ void fused_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned h = fused_pack_bf16x2(v[2 * q], v[2 * q + 1]);
    hi[q] = h;
    lo[q] = fused_pack_bf16x2(v[2 * q] - __uint_as_float(h << 16), v[2 * q + 1] - __uint_as_float(h & 0xffff0000u));
  }
}
// lane n of a 16-lane row receives lane n - 1 (shr) / n + 1 (shl); lanes without a source receive 0
__device__ __forceinline__ unsigned fused_row_shr1(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned fused_row_shl1(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true); }

#ifdef V3D_PHASE_TIMING
// developer build only: wave 0 of every workgroup adds up the cycles between marks: 0 wait for the slab (vmcnt), 1 second half of the
// layer-1 steps, 2 head, 3 layer transitions, 4 / 5 second half of the layer 2 / 3 steps, 6 first half of a step, 7 waits at the step barrier
constexpr int kFPhaseSlots = 1 << 12;
__device__ unsigned long long g_fused_phase[8 * kFPhaseSlots];
#define FPHASE_DECL long long ph_t = __builtin_readcyclecounter(); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FPHASE_MARK(i) do { long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#define FPHASE_FLUSH do { if (threadIdx.x == 0 && blockIdx.x < kFPhaseSlots) for (int i_ = 0; i_ < 8; ++i_) g_fused_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_]; } while (0)
#else
#define FPHASE_DECL
#define FPHASE_MARK(i)
#define FPHASE_FLUSH
#endif
#ifndef V3D_FUSED_FETCH64
#define V3D_FUSED_FETCH64 0       // developer experiment: 1 = the staged B fragment is fetched with two 16-byte reads (the hazard)
#endif
#ifndef V3D_FUSED_SAFE_WAIT
#define V3D_FUSED_SAFE_WAIT 0     // developer experiment: every rendezvous drains the wave's vector-memory queue
#endif
#ifndef V3D_FUSED_ABLATE
#define V3D_FUSED_ABLATE 0   // developer ablations: 1 no matrix instructions, 2 gathers from rows 0..7 only, 3 no weight DMA, 6 no gathers, 8 neither
#endif

__global__ __launch_bounds__(kDThreads, 2) void decoder_fused_kernel(FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const cbias = reinterpret_cast<float*>(smem + kDLdsRing + kDLdsCtab + kDLdsStage);   // [3 layers][4 mb][2 g][16 regs]
  float* const chead = cbias + 3 * kDH;                                               // [3 taps][4 mb][2 g][16 regs]
  float* const cmisc = chead + 3 * kDH;                                               // head bias, -, -, -, offset values [8]
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x2* const ctab = reinterpret_cast<u32x2*>(smem + kDLdsRing + wave * kDCtabWave);   // wave-private [4][8][32]
  const unsigned* const ctab32 = reinterpret_cast<const unsigned*>(ctab);
  unsigned char* const stage = smem + kDLdsRing + kDLdsCtab + wave * kDStageWave;       // wave-private [hi, lo][g][column] 16 B
  const int n_hyp = p.n_hyp;
  const int sb1 = p.C[0] >> 4, sb2 = sb1 + (p.C[1] >> 4), sb3 = sb2 + (p.C[2] >> 4), n1 = p.nstep1;   // step ranges of layer 1
  FPHASE_DECL;

  // tile-invariant constants, parked in LDS in the register order of the accumulator tile: entry [mb][g][r] belongs to channel
  // 32 mb + 8 (r >> 2) + 4 g + (r & 3)
  for (int i = tid; i < 3 * kDH; i += kDThreads) {
    const int L = i >> 7, j = i & 127, mb = j >> 5, gg = (j >> 4) & 1, r = j & 15;
    const int ch = 32 * mb + 8 * (r >> 2) + 4 * gg + (r & 3);
    cbias[i] = (L == 0 ? p.bias[0] : L == 1 ? p.bias[1] : p.bias[2])[ch];
    chead[i] = p.head_w[ch * 3 + L];                          // weight [1, C, 3]: L = tap
  }
  if (tid == 0) cmisc[0] = p.head_b[0];
  if (tid < 8) cmisc[4 + tid] = (p.vals && tid < n_hyp) ? p.vals[tid] : 0.f;

  // Tile walk: workgroups are dealt round-robin to the 8 XCDs, so XCD x = blockIdx.x % 8 takes the x-th contiguous eighth of the
  // tiles (consecutive tiles are neighbouring pixels of one view): the corner rows its workgroups gather then come from the part
  // of the scene a few views see and stay in that XCD's 4 MB L2, instead of every XCD streaming all three feature tables.
  const int n_tiles = (p.n_pts + kDPtsTile - 1) / kDPtsTile;
  const int n_xcd = gridDim.x % 8 == 0 ? 8 : 1;
  const int tiles_per_xcd = (n_tiles + n_xcd - 1) / n_xcd, tile_step = gridDim.x / n_xcd;
  const int tile_end = min(n_tiles, ((int)blockIdx.x % n_xcd + 1) * tiles_per_xcd);
  int tile = ((int)blockIdx.x % n_xcd) * tiles_per_xcd + (int)blockIdx.x / n_xcd;
  if (tile >= tile_end) return;

  // ---- weight ring ------------------------------------------------------------------------------------------------------------
  // this wave's three 1 KB pieces of the slab at `base` -> ring slot `slot` (hand-written LDS-DMA: M0 carries the LDS address, the
  // global address is an SGPR base + lane * 16)
  auto dma_issue = [&](const char* base, int slot) __attribute__((always_inline)) {
    if (V3D_FUSED_ABLATE == 3 || V3D_FUSED_ABLATE == 9 || V3D_FUSED_ABLATE == 12) return;
    base += wave * (kDPieces * 1024);
    const unsigned dst = ring_lds + (unsigned)slot * kDSlab + (unsigned)wave * (kDPieces * 1024);
    const unsigned voff = (unsigned)(threadIdx.x & 63) * 16u;
    const char* const b1 = base + 1024;
    const char* const b2 = base + 2048;
    unsigned m0v;
    asm volatile(
        "s_mov_b32 %[m0v], m0\n\t"
        "s_mov_b32 m0, %[dst]\n\t"
        "global_load_lds_dwordx4 %[v], %[b0]\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %[v], %[b1]\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %[v], %[b2]\n\t"
        "s_mov_b32 m0, %[m0v]"
        : [m0v] "=&s"(m0v)
        : [dst] "s"(dst), [b0] "s"(base), [b1] "s"(b1), [b2] "s"(b2), [v] "v"(voff)
        : "memory", "scc");
  };
  const char* const w0 = reinterpret_cast<const char*>(p.w[0]);
  const char* const w1 = reinterpret_cast<const char*>(p.w[1]);
  const char* const w2 = reinterpret_cast<const char*>(p.w[2]);
  // slab of layer-1 step `st`; st = n1, n1 + 1 are the first two steps of the second layer
  auto slab1 = [&](int st) __attribute__((always_inline)) { return (st < n1 ? w0 : w1) + (size_t)(st < n1 ? st : st - n1) * kDSlab; };
  int slot = 0;            // ring slot of the current step
  // The one rendezvous of a step sits in its MIDDLE (between matrix-instruction groups 2 and 3), so that no wave ever starts a step
  // with empty hands: the slab of step s + 1 was requested in the middle of step s - 1; in the middle of step s this wave's pieces of
  // it must have landed (WAIT = the number of vector-memory operations the wave has issued since that request: the counter retires
  // in order), the barrier publishes it -- from group 5 on the wave prefetches the first A fragments of step s + 1 -- and says that
  // every wave has left step s - 1, whose slot takes the slab of step s + 2 (`next2`; behind the last tile the sequence simply wraps
  // around: the surplus slabs are never read).
  auto mid_sync = [&](const char* next2, auto wait) __attribute__((always_inline)) {
    constexpr int W = decltype(wait)::value;
    FPHASE_MARK(6);
    if constexpr (W == 0 || V3D_FUSED_SAFE_WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (W == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (W == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(W == 0, "mid_sync: unsupported wait");
    FPHASE_MARK(0);
#if V3D_FUSED_ABLATE != 11 && V3D_FUSED_ABLATE != 12        // (11: timing experiment without the rendezvous -- results are wrong)
    __syncthreads();
#endif
    FPHASE_MARK(7);
    dma_issue(next2, slot == 0 ? 2 : slot - 1);
  };
  auto step_done = [&]() __attribute__((always_inline)) { slot = slot == 2 ? 0 : slot + 1; };
  using W0 = std::integral_constant<int, 0>;
  using W12 = std::integral_constant<int, 12>;
  using W16 = std::integral_constant<int, 16>;

  // ---- per-lane geometry: re-derived from an opaque copy of threadIdx.x in every tile so that nothing per-lane is hoisted out of
  // (and kept alive across) the tile loop
  int lane, g, n, h;                         // consumer view: MFMA column n = lane & 31, k half g = lane >> 5
  int pcol, ppc;                             // producer view: quad lane >> 2 owns columns pcol and pcol + 16, lane & 3 = 16-byte piece
  unsigned m_t0, m_t2;                       // tap 0 (h - 1) / tap 2 (h + 1) stays inside the hypothesis group
  auto refresh_lane_ids = [&]() __attribute__((always_inline)) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    tid = t; lane = t & 63; g = lane >> 5; n = lane & 31; h = n & 7;
    pcol = lane >> 2; ppc = lane & 3;
    m_t0 = h >= 1 ? 0xffffffffu : 0u;
    m_t2 = h + 1 < n_hyp ? 0xffffffffu : 0u;
  };
  refresh_lane_ids();

  // The tile's corner table, wave-private in LDS: [level 0..2 | point features][corner][column] (row, weight).  Lane (g, n)
  // fetches corners 4 g .. 4 g + 3 of the three levels for column n from decoder_corner_kernel's table; the pseudo-level 3 makes the
  // per-point feature channels of the first layer one more "interpolation": corner 0 = (row of the hypothesis point, 1.0),
  // corners 1..7 = (0, 0.0).  Invalid columns (hypothesis slot >= n_hyp, point >= n_pts) read row 0 with weight 0 everywhere.
  auto ctab_load = [&](u32x4 (&ce)[3][2], int tl) __attribute__((always_inline)) {
    const int pt = tl * kDPtsTile + wave * kDPtsWave + (n >> 3);
    const bool ok = pt < p.n_pts && h < n_hyp;
    // (unconditional loads from a clamped address, masked afterwards: a select makes the compiler branch around each load and
    // wait for it)
    const unsigned okm = ok ? 0xffffffffu : 0u;
    const u32x4* src = reinterpret_cast<const u32x4*>(p.ctab + ((size_t)(ok ? pt : 0) * n_hyp + (ok ? h : 0)) * 24 + 4 * g);
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j) ce[l][j] = src[l * 4 + j] & (u32x4){okm, okm, okm, okm};
  };
  auto ctab_store = [&](const u32x4 (&ce)[3][2], int tl) __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ctab[(l * 8 + 4 * g + 2 * j) * 32 + n] = (u32x2){ce[l][j][0], ce[l][j][1]};
        ctab[(l * 8 + 4 * g + 2 * j + 1) * 32 + n] = (u32x2){ce[l][j][2], ce[l][j][3]};
      }
    const int pt = tl * kDPtsTile + wave * kDPtsWave + (n >> 3);
    const bool ok = pt < p.n_pts && h < n_hyp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * g + j;
      ctab[(24 + k) * 32 + n] = (k == 0 && ok) ? (u32x2){(unsigned)(pt * n_hyp + h), __float_as_uint(1.f)} : (u32x2){0u, 0u};
    }
  };

  // ---- layer-1 producer ----------------------------------------------------------------------------------------------------------
  // A quad (4 lanes x 16 bytes = 64 contiguous bytes of one feature row: the 16 channels of a K step) per column and load
  // instruction -- a quad that reads four different rows costs the CU's one vector-memory path four times as much -- two columns
  // per quad (pcol, pcol + 16), 8 corner rows each: 16 gathers per step and lane, blended in the lane that loaded them, split,
  // and handed to the consumer's lane layout through a 2 KB wave-private LDS tile.  The gathers form a rotating pipeline one step
  // deep: in step s, between the matrix instructions, corner k of step s + 1's rows (requested during step s - 1) is consumed and
  // the same registers immediately request corner k of step s + 2 -- every load has a whole step to land, the loads of a wave
  // are spread over the step, and 64 registers hold them.
  struct StepSrc { const char* base; unsigned rowb, cofs; int ct; };            // wave-uniform description of a layer-1 step
  auto src_of = [&](int u) __attribute__((always_inline)) {
    StepSrc q;
    const int l = u < sb1 ? 0 : u < sb2 ? 1 : u < sb3 ? 2 : 3;
    q.base = reinterpret_cast<const char*>(l == 0 ? p.feats[0] : l == 1 ? p.feats[1] : l == 2 ? p.feats[2] : p.pts_feat);
    q.rowb = (unsigned)(l == 0 ? p.C[0] : l == 1 ? p.C[1] : l == 2 ? p.C[2] : p.c_feat) * 4u;
    q.cofs = (unsigned)(u - (l == 0 ? 0 : l == 1 ? sb1 : l == 2 ? sb2 : sb3)) * 64u;
    q.ct = l * 256;
    return q;
  };
  f32x4 gx[8][2];          // [corner][column half]
  f32x4 pa0, pa1;          // the two columns' blends in progress
  // (rows and row pitches fit 24 bits -- checked on the host -- so the product is a full-rate v_mad_u32_u24)
  auto prod_issue_row = [&](int k, int c, unsigned row, const StepSrc& q) __attribute__((always_inline)) {
    const unsigned off = __umul24(V3D_FUSED_ABLATE == 2 ? (row & 7u) : row, q.rowb) + q.cofs + (unsigned)ppc * 16u;
#if V3D_FUSED_ABLATE == 6 || V3D_FUSED_ABLATE == 8 || V3D_FUSED_ABLATE == 9 || V3D_FUSED_ABLATE == 10   // timing experiment: no gathers
    gx[k][c] = (f32x4){__uint_as_float(off), 0.f, 0.f, 0.f};
#else
    gx[k][c] = *reinterpret_cast<const f32x4*>(q.base + (size_t)off);
#endif
  };
  auto prod_issue = [&](int k, const StepSrc& q) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) prod_issue_row(k, c, ctab32[2 * (q.ct + k * 32 + pcol + 16 * c)], q);
  };
  // corner order x fastest; absent corners add nothing (no renormalisation)
  auto prod_consume_w = [&](int k, float wa, float wb) __attribute__((always_inline)) {
    pa0 = __builtin_elementwise_fma(gx[k][0], (f32x4){wa, wa, wa, wa}, pa0);
    pa1 = __builtin_elementwise_fma(gx[k][1], (f32x4){wb, wb, wb, wb}, pa1);
  };
  auto prod_consume = [&](int k, const StepSrc& q) __attribute__((always_inline)) {
    prod_consume_w(k, __uint_as_float(ctab32[2 * (q.ct + k * 32 + pcol) + 1]), __uint_as_float(ctab32[2 * (q.ct + k * 32 + pcol + 16) + 1]));
  };
  // the finished blends -> split -> the consumer layout: piece ppc = 2 j + g' holds the channels 8 j + 4 g' .. + 3 of the step, i.e.
  // half j of the B fragment of lane (g', column)
  auto prod_finalize = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x4 a = c == 0 ? pa0 : pa1;
      const unsigned h01 = fused_pack_bf16x2(a.x, a.y), h23 = fused_pack_bf16x2(a.z, a.w);
      const unsigned l01 = fused_pack_bf16x2(a.x - __uint_as_float(h01 << 16), a.y - __uint_as_float(h01 & 0xffff0000u));
      const unsigned l23 = fused_pack_bf16x2(a.z - __uint_as_float(h23 << 16), a.w - __uint_as_float(h23 & 0xffff0000u));
      unsigned char* const d = stage + (((ppc & 1) * 32 + pcol + 16 * c) * 16 + (ppc >> 1) * 8);
      *reinterpret_cast<u32x2*>(d) = (u32x2){h01, h23};
      *reinterpret_cast<u32x2*>(d + 1024) = (u32x2){l01, l23};
    }
    pa0 = pa1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto prod_fetch = [&](u32x4& xh, u32x4& xl) __attribute__((always_inline)) {
#if V3D_FUSED_FETCH64 == 0
    // 8-byte reads, waited for in place: a 16-byte LDS read whose result feeds VECTOR instructions (here the DPP shifts of the next
    // step) has returned the registers' previous contents in some lanes when matrix instructions were in flight on the SIMD (the
    // round-3 hazard of the corner table; reproduced in round 5 with this very read: a few points per launch kept
    // the previous step's B fragment); the A fragments -- 16-byte reads consumed by matrix instructions -- are not affected.
    const unsigned sa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)stage + (unsigned)lane * 16u;
    u32x2 h0, h1, l0, l1;
    asm volatile(
        "ds_read_b64 %0, %4\n\t"
        "ds_read_b64 %1, %4 offset:8\n\t"
        "ds_read_b64 %2, %4 offset:1024\n\t"
        "ds_read_b64 %3, %4 offset:1032\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)
        : "v"(sa)
        : "memory");
    xh = (u32x4){h0[0], h0[1], h1[0], h1[1]};
    xl = (u32x4){l0[0], l0[1], l1[0], l1[1]};
#else
    xh = *reinterpret_cast<const u32x4*>(stage + lane * 16);
    xl = *reinterpret_cast<const u32x4*>(stage + 1024 + lane * 16);
#endif
  };

  // ---- one K step: 3 taps x 4 row blocks x (hi*hi, hi*lo, lo*hi) on the slab in ring slot `slot`, as six groups of six matrix
  // instructions; hook(i) runs behind group i (the producer's share of the step)
  f32x16 acc[4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
  };
  // A fragments of group (t, mp): row blocks 2 mp, 2 mp + 1, hi and lo.  The next group's are requested before this group's matrix
  // instructions (an LDS round trip per group in front of them otherwise); group 5 requests group 0 of the NEXT step's slab,
  // published by this step's rendezvous.
  u32x4 af[2][4];
  auto load_a = [&](u32x4 (&a)[4], int sl, int t, int mp, const u32x4 xh, const u32x4 xl) __attribute__((always_inline)) {
    const u32x4* const slab = reinterpret_cast<const u32x4*>(smem + sl * kDSlab) + lane;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#if V3D_FUSED_ABLATE == 10        // timing experiment: no A-fragment reads either
      a[j] = xh; a[2 + j] = xl;
#else
      a[j] = slab[((t * 2 + 0) * 4 + 2 * mp + j) * 64];
      a[2 + j] = slab[((t * 2 + 1) * 4 + 2 * mp + j) * 64];
#endif
    }
  };
  auto mma_step = [&](const u32x4 xh, const u32x4 xl, auto hook, auto mid) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      u32x4 th = xh, tl = xl;
      if (t != 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          th[q] = (t == 0 ? fused_row_shr1(xh[q]) : fused_row_shl1(xh[q])) & (t == 0 ? m_t0 : m_t2);
          tl[q] = (t == 0 ? fused_row_shr1(xl[q]) : fused_row_shl1(xl[q])) & (t == 0 ? m_t0 : m_t2);
        }
      }
      const bf16x8 b_hi = __builtin_bit_cast(bf16x8, th), b_lo = __builtin_bit_cast(bf16x8, tl);
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {
        const int grp = t * 2 + mp;
        if (grp < 5) load_a(af[(grp + 1) & 1], slot, (grp + 1) >> 1, (grp + 1) & 1, xh, xl);
        else load_a(af[0], slot == 2 ? 0 : slot + 1, 0, 0, xh, xl);
        hook(std::integral_constant<int, 1>{}, grp);      // (the producer's table reads of this group, too)
        __builtin_amdgcn_sched_barrier(0);     // (the requests stay in front of the matrix instructions)
        const u32x4 (&a)[4] = af[grp & 1];
#if V3D_FUSED_ABLATE == 1 || V3D_FUSED_ABLATE == 8 || V3D_FUSED_ABLATE == 9 || V3D_FUSED_ABLATE == 10
        asm volatile("" : : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b_hi), "v"(b_lo));
#else
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[2 * mp + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), b_hi, acc[2 * mp + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[2 * mp + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), b_lo, acc[2 * mp + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[2 * mp + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 + j]), b_hi, acc[2 * mp + j], 0, 0, 0);
#endif
        hook(std::integral_constant<int, 0>{}, grp);
        if (grp == 2) mid();
        __builtin_amdgcn_sched_barrier(0);     // the producer's loads stay where they are written: spread over the step
      }
    }
  };
  auto no_hook = [](auto, int) __attribute__((always_inline)) {};      // hook(phase, group): phase 1 in front of the group's matrix instructions, 0 behind
  // the producer's share of a step: groups 0..3 turn over two corners each (consume step `cu`'s rows, request step `iu`'s), group
  // 3 also stages the finished B fragment, group 4 fetches it in the consumer layout
  u32x4 bh, bl;            // B fragment (hi, lo) of the layer-1 step about to run
  u32x4 bhn, bln;
  // The table entries a group's share needs -- weights of the rows it consumes, row indices of those it requests -- are read in
  // front of the group's matrix instructions: behind them they were two LDS round trips in a row per group, in the issue path of
  // the wave's next matrix instructions.
  unsigned pr[4];
  float pw[4];
  auto prod_hook = [&](const StepSrc& cs, const StepSrc& is, bool consume) __attribute__((always_inline)) {
    return [&, consume](auto phase, int grp) __attribute__((always_inline)) {
      if constexpr (decltype(phase)::value == 1) {
        if (grp < 4) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int k = 2 * grp + kk;
              // (4-byte reads on purpose: see prod_fetch -- LDS reads that return 16 bytes per lane must not feed vector instructions)
              if (consume) pw[2 * kk + c] = __uint_as_float(ctab32[2 * (cs.ct + k * 32 + pcol + 16 * c) + 1]);
              pr[2 * kk + c] = ctab32[2 * (is.ct + k * 32 + pcol + 16 * c)];
            }
        }
      } else {
        if (grp < 4) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * grp + kk;
            if (consume) prod_consume_w(k, pw[2 * kk], pw[2 * kk + 1]);
            prod_issue_row(k, 0, pr[2 * kk], is);
            prod_issue_row(k, 1, pr[2 * kk + 1], is);
          }
          if (grp == 3 && consume) prod_finalize();
        } else if (grp == 4) {
          if (consume) prod_fetch(bhn, bln);
        }
      }
    };
  };
  // bias + ReLU of the accumulator tile -> the 8 B fragments (hi, lo) of the next 128 -> 128 layer
  u32x4 ah[8], al[8];
  auto next_layer_operands = [&](int layer) __attribute__((always_inline)) {
    const float* const bs = cbias + layer * kDH + g * 16;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float bq[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + mb * 32 + q * 4);
        bq[4 * q] = b4.x; bq[4 * q + 1] = b4.y; bq[4 * q + 2] = b4.z; bq[4 * q + 3] = b4.w;
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(acc[mb][8 * pp + e] + bq[8 * pp + e], 0.f);
        fused_split8(v, ah[2 * mb + pp], al[2 * mb + pp]);
      }
    }
  };

  // ---- prologue: the first two slabs, the first tile's corner table, the producer pipeline primed for steps 0 and 1 --------------
  dma_issue(slab1(0), 0);
  dma_issue(slab1(1), 1);
  {
    u32x4 ce[3][2];
    ctab_load(ce, tile);
    ctab_store(ce, tile);
  }
  pa0 = pa1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const StepSrc s0 = src_of(0), s1 = src_of(min(1, n1 - 1));
#pragma unroll
    for (int k = 0; k < 8; ++k) prod_issue(k, s0);
#pragma unroll
    for (int k = 0; k < 8; ++k) { prod_consume(k, s0); prod_issue(k, s1); }
    prod_finalize();
    prod_fetch(bh, bl);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // both slabs (and the rows of step 1) have landed
  __syncthreads();                                      // ... in every wave; cbias / chead / cmisc visible
  load_a(af[0], 0, 0, 0, bh, bl);

#pragma unroll 1
  for (; tile < tile_end; tile += tile_step) {
    refresh_lane_ids();
    const bool has_next = tile + tile_step < tile_end;
    zero_acc();
    FPHASE_MARK(3);

    // ---- layer 1: every step consumes the rows of the next step and requests those of the step after next (clamped to the last
    // step at the end: the surplus B fragments are never used) -- 16 gathers per step, hence the 16 of the rendezvous ------------
#pragma unroll 1
    for (int s = 0; s < n1; ++s) {
      const StepSrc cs = src_of(min(s + 1, n1 - 1)), is = src_of(min(s + 2, n1 - 1));
      const char* const nx = slab1(s + 2);
      mma_step(bh, bl, prod_hook(cs, is, true), [&]() __attribute__((always_inline)) { mid_sync(nx, W16{}); });
      bh = bhn; bl = bln;
      step_done();
      FPHASE_MARK(1);
    }

    // ---- layers 2 and 3: B fragments straight from the accumulators; the next tile's corner table is fetched on the side (late in
    // layer 2, when half of its B fragments are dead) -----------------------------------------------------------------------------
    next_layer_operands(0);
    zero_acc();
    FPHASE_MARK(3);
    {
      u32x4 ce[3][2];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const char* const nx = s + 2 < 8 ? w1 + (size_t)(s + 2) * kDSlab : w2 + (size_t)(s + 2 - 8) * kDSlab;
        if (s == 4 && has_next) ctab_load(ce, tile + tile_step);
        mma_step(ah[s], al[s], no_hook, [&]() __attribute__((always_inline)) { mid_sync(nx, W0{}); });
        if (s == 5 && has_next) ctab_store(ce, tile + tile_step);       // (fetched during step 4: landed, by the vmcnt(0) of step 5)
        step_done();
        FPHASE_MARK(4);
      }
    }
    next_layer_operands(1);
    zero_acc();
    FPHASE_MARK(3);
    {
      // steps 6 and 7 prime the producer for the next tile: step 6 requests the rows of its step 0, step 7 consumes them, requests
      // those of its step 1 and leaves the B fragment of its step 0 in (bh, bl)
      const StepSrc s0 = src_of(0), s1 = src_of(min(1, n1 - 1));
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const char* const nx = s + 2 < 8 ? w2 + (size_t)(s + 2) * kDSlab : w0 + (size_t)(s + 2 - 8) * kDSlab;
        if (s == 6) mma_step(ah[s], al[s], prod_hook(s0, s0, false), [&]() __attribute__((always_inline)) { mid_sync(nx, W12{}); });
        else if (s == 7) mma_step(ah[s], al[s], prod_hook(s0, s1, true), [&]() __attribute__((always_inline)) { mid_sync(nx, W16{}); });
        else mma_step(ah[s], al[s], no_hook, [&]() __attribute__((always_inline)) { mid_sync(nx, W0{}); });
        step_done();
        FPHASE_MARK(5);
      }
      bh = bhn; bl = bln;
    }

    // ---- head: Conv1d(128 -> 1, k3, pad 1, bias) over the hypotheses, softmax, expectation ------------------------------------
    {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      const float* const bs = cbias + 2 * kDH + g * 16;
      const float* const hw = chead + g * 16;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + mb * 32 + q * 4);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(hw + mb * 32 + q * 4);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(hw + kDH + mb * 32 + q * 4);
          const f32x4 w2 = *reinterpret_cast<const f32x4*>(hw + 2 * kDH + mb * 32 + q * 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y = fmaxf(acc[mb][4 * q + k] + b4[k], 0.f);
            s0 = fmaf(y, w0[k], s0); s1 = fmaf(y, w1[k], s1); s2 = fmaf(y, w2[k], s2);
          }
        }
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      // out[h] = in[h - 1] w[0] + in[h] w[1] + in[h + 1] w[2]
      float sc = s1 + __uint_as_float(fused_row_shr1(__float_as_uint(s0)) & m_t0) +
                 __uint_as_float(fused_row_shl1(__float_as_uint(s2)) & m_t2) + cmisc[0];
      const bool hyp_ok = h < n_hyp;
      if (!hyp_ok) sc = -INFINITY;
      float m = sc;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
      const float ex = hyp_ok ? expf(sc - m) : 0.f;
      float sum = ex;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
      const float pr = ex / sum;
      float e = cmisc[4 + h] * pr;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) e += __shfl_xor(e, o);
      const int pt = tile * kDPtsTile + wave * kDPtsWave + (n >> 3);
      if (g == 0 && pt < p.n_pts) {
        if (hyp_ok) p.preds[(size_t)pt * n_hyp + h] = pr;
        if (h == 0 && p.expect) p.expect[pt] = e;
        if (h == 0 && p.depth_io) p.depth_io[pt] += e;
      }
    }
    FPHASE_MARK(2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may be in flight when the workgroup's LDS is released
  FPHASE_FLUSH;
}
}  // namespace

extern "C" int v3d_decoder_head_f32(const float* act, int n_pts, int n_hyp, int C, const float* weight,
                                    const float* bias, const float* offset_vals, float* preds,
                                    float* expect, void* stream) {
  V3D_REQUIRE(act && weight && bias && preds, V3D_ERR_BAD_ARG, "v3d_decoder_head_f32: null argument");
  V3D_REQUIRE(n_hyp >= 1 && n_hyp <= kMaxHyp && C >= 1 && n_pts >= 0, V3D_ERR_BAD_SHAPE,
              "v3d_decoder_head_f32: bad shape (n_hyp=%d)", n_hyp);
  V3D_REQUIRE(!expect || offset_vals, V3D_ERR_BAD_ARG, "v3d_decoder_head_f32: expect without offset_vals");
  if (n_pts == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  v3d::TimedScope ts("decoder_head", s);
  decoder_head_kernel<<<(n_pts + 3) / 4, 256, 0, s>>>(act, n_pts, n_hyp, C, weight, bias, offset_vals, preds, expect);
  V3D_CHECK_LAUNCH("decoder_head_kernel");
  return V3D_OK;
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_fused_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)8 * kFPhaseSlots);
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fused_phase), h.size() * sizeof(unsigned long long)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < kFPhaseSlots; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif

#ifdef V3D_CORNER_STATS
extern "C" int v3d_debug_corner_stats(unsigned long long* out12, int reset) {
  if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_corner_stats), sizeof(unsigned long long) * 12) != hipSuccess) return 1;
  if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_corner_stats), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

extern "C" size_t v3d_decoder_fused_workspace_bytes(int n_pts, int n_hyp) {
  if (n_pts <= 0 || n_hyp <= 0) return 256;
  return (size_t)n_pts * (size_t)n_hyp * 24 * sizeof(u32x2) + 256;
}

extern "C" int v3d_decoder_fused_f32(const v3d_gemm_weights* const* layers_host, const float* head_weight,
                                     const float* head_bias, const void* const* level_table_host,
                                     const int* level_n_host, const float* const* level_feats_host,
                                     const int* level_C_host, const int* level_stride_host,
                                     const float* const* level_min_pts_host, const float* level_res_host,
                                     const float* pts, const int64_t* pts_batch, const float* pts_feat, int c_feat,
                                     int n_pts, int n_hyp, const float* offset_vals, float* preds, float* expect,
                                     float* depth_inout, void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(layers_host && head_weight && head_bias && level_table_host && level_n_host && level_feats_host &&
                  level_C_host && level_stride_host && level_min_pts_host && level_res_host && pts && pts_batch && preds,
              V3D_ERR_BAD_ARG, "v3d_decoder_fused_f32: null argument");
  V3D_REQUIRE(n_pts >= 0 && n_hyp >= 1 && n_hyp <= 8, V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: n_hyp=%d (1..8)", n_hyp);
  // (24-bit row indices in the kernel's gather addresses)
  V3D_REQUIRE((long long)n_pts * n_hyp < (1ll << 24), V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: n_pts=%d too large for one call", n_pts);
  V3D_REQUIRE(c_feat >= 0 && c_feat % 16 == 0 && (c_feat == 0 || pts_feat), V3D_ERR_UNSUPPORTED,
              "v3d_decoder_fused_f32: c_feat=%d must be a multiple of 16 (with pts_feat given)", c_feat);
  V3D_REQUIRE((!expect && !depth_inout) || offset_vals, V3D_ERR_BAD_ARG, "v3d_decoder_fused_f32: expect / depth_inout without offset_vals");
  V3D_REQUIRE(workspace && workspace_bytes >= v3d_decoder_fused_workspace_bytes(n_pts, n_hyp), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_decoder_fused_f32: workspace of %zu bytes, need %zu", workspace_bytes,
              v3d_decoder_fused_workspace_bytes(n_pts, n_hyp));
  V3D_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0 && (!pts_feat || (reinterpret_cast<size_t>(pts_feat) & 15) == 0),
              V3D_ERR_BAD_ARG, "v3d_decoder_fused_f32: workspace / pts_feat must be 16-byte aligned");
  FusedParams p;
  CornerParams cp;
  memset(&p, 0, sizeof(p));
  memset(&cp, 0, sizeof(cp));
  int k1 = c_feat;
  bool pow2 = true;
  for (int l = 0; l < 3; ++l) {
    V3D_REQUIRE(level_table_host[l] && level_feats_host[l] && level_min_pts_host[l] && level_n_host[l] > 0 &&
                    level_C_host[l] > 0 && level_C_host[l] % 16 == 0 && level_stride_host[l] > 0 && level_res_host[l] > 0.f,
                V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: level %d (channels must be a multiple of 16)", l);
    V3D_REQUIRE((reinterpret_cast<size_t>(level_feats_host[l]) & 15) == 0, V3D_ERR_BAD_ARG,
                "v3d_decoder_fused_f32: level %d features must be 16-byte aligned", l);
    // (the kernel addresses a level's rows with 32-bit byte offsets)
    V3D_REQUIRE((long long)level_n_host[l] * level_C_host[l] * 4 < (1ll << 31) && level_n_host[l] < (1 << 24), V3D_ERR_UNSUPPORTED,
                "v3d_decoder_fused_f32: level %d holds %d x %d floats (2 GB limit)", l, level_n_host[l], level_C_host[l]);
    cp.table[l] = v3dhash::table_view(const_cast<void*>(level_table_host[l]), level_n_host[l]);
    cp.min_pts[l] = level_min_pts_host[l]; cp.res[l] = level_res_host[l];
    cp.tsf[l] = (float)level_stride_host[l]; cp.inv_ts[l] = 1.f / (float)level_stride_host[l];
    pow2 = pow2 && (level_stride_host[l] & (level_stride_host[l] - 1)) == 0;
    p.feats[l] = level_feats_host[l]; p.C[l] = level_C_host[l];
    k1 += level_C_host[l];
  }
  for (int l = 0; l < 3; ++l) {
    const v3d_gemm_weights* h = layers_host[l];
    V3D_REQUIRE(h && h->n_seg == 3 && h->N == kDH && h->has_bias && h->K == (l == 0 ? k1 : kDH) && h->dec_ofs != 0,
                V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: layer %d must be a packed Conv1d(k3) %d -> 128 with bias", l,
                l == 0 ? k1 : kDH);
    p.w[l] = h->dev + h->dec_ofs;
    p.bias[l] = h->dev + h->bias_ofs;
  }
  p.nstep1 = k1 / 16;
  V3D_REQUIRE(p.nstep1 >= 2, V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: the first layer needs at least 32 input channels");
  p.ctab = reinterpret_cast<const u32x2*>(workspace);
  p.pts_feat = pts_feat; p.c_feat = c_feat; p.n_pts = n_pts; p.n_hyp = n_hyp;
  p.head_w = head_weight; p.head_b = head_bias; p.vals = offset_vals; p.preds = preds; p.expect = expect; p.depth_io = depth_inout;
  cp.pts = pts; cp.pts_batch = (const long long*)pts_batch; cp.n_hyp = n_hyp; cp.n_q = n_pts * n_hyp;
  cp.out = reinterpret_cast<u32x2*>(workspace);
  if (n_pts == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  // per device: the dynamic-LDS opt-in and the number of CUs (one workgroup each)
  static int resident_of[64] = {0};
  int dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: device ordinal %d", dev);
  if (!resident_of[dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)decoder_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDLdsBytes));
    int n_cu = 0;
    V3D_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    resident_of[dev] = n_cu > 0 ? n_cu : 256;
  }
  {
    v3d::TimedScope ts("decoder_corners", s);
    const long long n_thr = (long long)cp.n_q * 8;
    cp.m_hyp = v3d::magic_u32((unsigned long long)cp.n_q, (unsigned)n_hyp);
    if (pow2) decoder_corner_kernel<true><<<(unsigned)((n_thr + 255) / 256), 256, 0, s>>>(cp);
    else decoder_corner_kernel<false><<<(unsigned)((n_thr + 255) / 256), 256, 0, s>>>(cp);
  }
  V3D_CHECK_LAUNCH("decoder_corner_kernel");
  {
    // persistent tile walk: workgroup b takes tiles b, b + grid, ... of its XCD's share (equal work per tile)
    const int n_tiles = (n_pts + kDPtsTile - 1) / kDPtsTile;
    const int resident = resident_of[dev];
    v3d::TimedScope ts("decoder_fused", s);
    decoder_fused_kernel<<<n_tiles < resident ? n_tiles : resident, kDThreads, kDLdsBytes, s>>>(p);
  }
  V3D_CHECK_LAUNCH("decoder_fused_kernel");
  return V3D_OK;
}
