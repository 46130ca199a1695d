// Row C2b tail + C3 of SURVEY.md §8a: the hypothesis decoder's last Conv1d(h_dim -> 1, k3, pad 1, bias)
// along the hypothesis axis, softmax over the hypotheses (refinement.py:24,43) and, optionally, the
// expected depth offset sum_i p_i * vals_i (lightningmodel.py:238-241).  One wave per point.
#include <cstdlib>
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
// Fused hypothesis decoder (rows C2a + C2b + C3 in ONE kernel): sparse trilinear interpolation of
// the three U-Net levels (refinement.py:28-41) -> Conv1d+BN+ReLU x3 along the hypothesis axis (:16-23) -> Conv1d(128 -> 1)
// + softmax (:24,43) -> expected offset (lightningmodel.py:237-241).  The [Nq, 352, 7] feature tensor (30.9 MB per
// reference view and sweep) and the three [Nq*7, 128] activations never reach HBM.
//
// One workgroup (4 waves) = kFPts = 8 query points = kFPts * n_hyp (<= 64) GEMM columns; whole hypothesis groups, so no conv
// tap crosses the tile.  Layer 1 consumes its 352 input channels in 32-wide chunks that are PRODUCED on the fly: every
// thread blends the 8 corner rows (hash-probed once per tile into an LDS corner table) of its (row, 4 channels) and commits
// the split-bf16 values to the staging tile the MFMAs read; the next chunk's gathers are in flight during the MFMAs of the
// current one.  Layer outputs (bias + ReLU) are split and kept in LDS in B-fragment order for the next layer; the last
// layer's output stays in LDS as fp32 for the 128 -> 1 head, softmax and expectation.  Matrix arithmetic as everywhere on
// this path: split-bf16 operands (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16), fp32 accumulation; weights are the
// split images of v3d_gemm_pack (A fragments go from L2 straight to registers, one tap ahead).
// LDS: two activation buffers of 33 KB (the second doubles as staging tile + corner table during layer 1).
// STATUS (round 3): the default decoder whenever the configuration allows (HypothesisDecoder.can_fuse): two workgroups per CU,
// 3.1 ms per 64-view sweep against 3.7 ms for v3d_sparse_interp_f32 + v3d_gemm_gather_f32 x 3 + v3d_decoder_head_f32 (see
// kFLdsBytes below for the occupancy bug of round 2).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef V3D_FUSED_PTS
#define V3D_FUSED_PTS 8
#endif
constexpr int kFPts = V3D_FUSED_PTS;          // query points per workgroup: 8 (4 waves, two workgroups per CU; 3.0 ms per 64-view
                                              // sweep) or 16 (8 waves, one per CU: 3.6 ms -- the barriers span twice the waves)
constexpr int kFRows = kFPts * 8;             // MFMA columns per workgroup (kFPts * n_hyp <= kFRows)
#ifndef V3D_FUSED_NB
#define V3D_FUSED_NB 4
#endif
constexpr int kFNB = V3D_FUSED_NB;            // column blocks (16 columns) per wave: 4, or 2 (twice the waves at half the accumulators)
constexpr int kFWaves = 4 * (kFRows / 16 / kFNB);   // waves per workgroup: wave = (row quarter, column group): 32 channels x 16 kFNB columns
constexpr int kFThreads = 64 * kFWaves;
constexpr int kFStageRows = kFRows * 8 / kFThreads;     // staging rows per thread (8 threads x 4 channels per row and chunk)
constexpr int kFNBT = kFRows / 16;            // column blocks of the workgroup
constexpr int kFZero = kFRows;           // index of the all-zero activation row (conv padding)
constexpr int kFRT = kFZero + 1;         // rows per LDS activation array
constexpr int kFH = 128;            // hidden width of the decoder
constexpr int kFMBW = 2, kFMB = 4 * kFMBW;      // per wave: kFNB column blocks x 2 row blocks
constexpr int kFWslab = 4 * kFMBW * 16 * 32;          // packed floats per (tap, K chunk) of a layer's weight image
constexpr size_t kFActBytes = (size_t)2 * kFRT * 16 * 16;                // [hi, lo][65 rows][16 slots of 8 bf16]
constexpr size_t kFStageBytes = (size_t)2 * kFRT * 4 * 16;               // [hi, lo][65 rows][4 slots]: one 32-channel chunk
constexpr size_t kFConstBytes = (3 * 128 + 128 * 3 + 4 + 8) * 4;            // biases of the three layers, head weights, head bias, offset values
constexpr size_t kFUsedLdsBytes = 2 * kFActBytes + kFConstBytes;
// Requested LDS: 80 KB = two workgroups per CU (two waves per SIMD: the kernel's 186 VGPRs + 50 AGPRs allow exactly that).
//
// The round-2 nondeterminism at two workgroups per CU, diagnosed in round 3 (scripts/micro/fused_decoder_stress.py dumps the
// layer-1 input rows a workgroup commits and compares them with v3d_sparse_interp_f32):
//   * it is not a race in this source: wrong rows appear in ~10 % of the workgroups on EVERY launch, always rows 6, 7 (mod 8)
//     of a wave's staging rows -- lanes 48..63 -- in single channels, as partial corner sums; workgroups at LDS base 0 and at
//     base 80 KB are hit alike; drains (vmcnt(0) + s_sleep), extra barriers and padded register allocations change nothing;
//   * it needs matrix instructions in flight on the SIMD (a build that reads the B fragments but issues no MFMA is clean) and
//     two waves per SIMD (any build whose register count admits one wave per SIMD is clean);
//   * it follows ONE code-generation choice: at -O2 / -O3 hipcc merges the eight consecutive corner weights a thread reads
//     from the LDS corner table into two ds_read_b128 (register tuples v[66:69], v[134:137]: even, not 4-aligned); -O1 emits
//     ds_read2_b32 and is clean, and so is -O3 with those reads kept scalar -- 300 launches beside a GEMM on a second stream.
// So the corner table is laid out corner-major ([corner][level][row]): the eight values a thread needs are 768 bytes apart,
// every read is a ds_read_b32 / ds_read2st64_b32 with an immediate offset and nothing can be merged into a 16-byte read
// (keeping the row-major table and reading it through volatile pointers also works, but every volatile access carries a full
// s_waitcnt, which serialises the gathers: 3.7 instead of 3.1 ms per sweep).  The GPU suite runs the repeated-launch
// determinism check at this occupancy.
constexpr size_t kFLdsBytes = kFPts == 8 ? 80 * 1024 : kFUsedLdsBytes;      // 8 points: two workgroups per CU; 16: one (132 KB)
__device__ __forceinline__ int fused_corner_index(int r, int l, int corner) { return (corner * 3 + l) * kFRows + r; }
static_assert(2 * kFStageBytes + 2 * kFRows * 3 * 8 * 4 <= kFActBytes, "two staging tiles + corner table alias the second buffer");
static_assert((size_t)kFRows * kFH * 4 <= kFActBytes, "fp32 output of the last layer aliases the first buffer");

struct FusedLevel {
  v3dhash::HashTable table;
  const float* feats;     // [N, C]
  const float* min_pts;   // [n_batch, 3]
  float res;              // x.res of the level (= tensor_stride * voxel size)
  int C, ts;
};

struct FusedParams {
  FusedLevel lv[3];       // in feature-row order: channels [0, C0) = lv[0] (finest), then lv[1], lv[2], then pts_feat
  const float* pts;       // [n_pts, n_hyp, 3]
  const long long* pts_batch;
  const float* pts_feat;  // [n_pts, n_hyp, c_feat] or null
  int c_feat, n_pts, n_hyp, nkc1;
  const float* w[3];      // split-bf16 weight images of the three Conv1d layers
  const float* bias[3];   // folded BatchNorm biases [128]
  const float* head_w;    // [1, 128, 3]
  const float* head_b;
  const float* vals;      // [n_hyp] offset values or null
  float* preds;           // [n_pts, n_hyp]
  float* expect;          // [n_pts] or null
#ifdef V3D_FUSED_DEBUG
  float* dbg_x;           // developer build: [n_pts * n_hyp, K1] the interpolated layer-1 input as committed to the staging tile
  unsigned* dbg_wg;       // developer build: per workgroup [4]: HW_REG_LDS_ALLOC, HW_REG_HW_ID, XCC id, 0
#endif
};

__device__ __forceinline__ unsigned fused_pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
// x = hi + lo (hi = RNE_bf16(x), lo = RNE_bf16(x - hi)) for 4 values -> two words of hi, two of lo
__device__ __forceinline__ void fused_split4(const float (&v)[4], u32x2& hi, u32x2& lo) {
  const unsigned h01 = fused_pack_bf16x2(v[0], v[1]), h23 = fused_pack_bf16x2(v[2], v[3]);
  hi = (u32x2){h01, h23};
  lo = (u32x2){fused_pack_bf16x2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u)),
               fused_pack_bf16x2(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u))};
}

#ifndef V3D_FUSED_LB
#define V3D_FUSED_LB 2      // two waves per SIMD = two workgroups per CU: the register allocator must stay within 256
#endif
#ifdef V3D_PHASE_TIMING
// developer build only (as in costreg.hip): wave 0 of every workgroup adds up the cycles between marks and writes them to
// its own slot; phases: 0 corner table, 1 layer-1 commits (two barriers + blend + split), 2 layer-1 matrix phase, 3 activation
// stores, 4 layer 2, 5 layer 3, 6 head
constexpr int kFPhaseSlots = 1 << 15;
__device__ unsigned long long g_fused_phase[8 * kFPhaseSlots];
#define FPHASE_DECL long long ph_t = __builtin_readcyclecounter(); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FPHASE_MARK(i) do { long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#define FPHASE_FLUSH do { if (threadIdx.x == 0 && blockIdx.x < kFPhaseSlots) for (int i_ = 0; i_ < 8; ++i_) g_fused_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_]; } while (0)
#else
#define FPHASE_DECL
#define FPHASE_MARK(i)
#define FPHASE_FLUSH
#endif
// -DV3D_PHASE_TIMING=2: the coarse marks collapse into slot 0 and the layer-1 chunk loop is resolved instead: 1 gather issue,
// 2 / 3 / 4 the three taps, 5 commit (wait for the gathers, blend, split, LDS writes), 6 barrier
#if defined(V3D_PHASE_TIMING) && V3D_PHASE_TIMING == 2
#undef FPHASE_MARK
#define FPHASE_MARK(i) do { long long t_ = __builtin_readcyclecounter(); ph_acc[0] += t_ - ph_t; ph_t = t_; } while (0)
#define FPHASE_FINE(i) do { long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define FPHASE_FINE(i)
#endif
#ifndef V3D_FUSED_ABLATE
#define V3D_FUSED_ABLATE 0   // developer ablations (scripts/micro/fused_decoder_ablate.sh): 1 no matrix instructions, 2 no feature gathers
                             // (corner rows read as row 0 ... of a 4 KB window), 3 weight fragments from a 2 KB window, 4 no hash probes,
                             // 5 layer 1 only, 7 tiles interleaved over the XCDs, 8 a chunk's gathers in one burst
#endif
__global__ __launch_bounds__(kFThreads, V3D_FUSED_LB) void decoder_fused_kernel(FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* const actA = reinterpret_cast<u32x4*>(smem);                               // [2][kFRT][16]
  u32x4* const actB = reinterpret_cast<u32x4*>(smem + kFActBytes);                  // [2][kFRT][16]
  u32x4* const xq = actB;                                                           // layer 1: two staging tiles [2][kFRT][4]
  int* const crow = reinterpret_cast<int*>(smem + kFActBytes + 2 * kFStageBytes);   // layer 1: [8 corners][3 levels][rows]
  float* const cw = reinterpret_cast<float*>(crow + kFRows * 24);
  constexpr int kFStageSlots = 2 * kFRT * 4;                                        // 16-byte slots of one staging tile
  float* const cbias = reinterpret_cast<float*>(smem + 2 * kFActBytes);             // [3][128] folded BatchNorm biases
  float* const chead = cbias + 3 * kFH;                                             // [128][3] head weights, then the head bias

  // Per-lane indices.  They are re-derived from an opaque copy of threadIdx.x at the top of every tile (refresh_lane_ids): with
  // plain loop invariants the compiler hoists every per-lane LDS / global address of every phase out of the tile loop and
  // keeps them all alive across it (256 VGPRs + 144 AGPRs + scratch instead of ~220 registers).
  int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_id & 3;            // row quarter: output channels 32 wave .. 32 wave + 31
  const int cb0 = (wave_id >> 2) * kFNB;   // first of this wave's 4 column blocks
  int kq = lane >> 4, jn = lane & 15;
  // query points per tile: as many whole hypothesis groups as the kFRows columns hold (7 hypotheses: 9 points = 63 of 64
  // columns; with a fixed 8 points per tile an eighth of every MFMA's columns was padding)
  const int n_hyp = p.n_hyp, npt = kFRows / n_hyp, rows = npt * n_hyp;
  const long long n_q = (long long)p.n_pts * n_hyp;
  const int n_tiles = (p.n_pts + npt - 1) / npt;
  // The workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the host launches as many workgroups as the chip holds at
  // once): everything that does not depend on the tile -- biases, head weights, the weight-fragment ring -- is set up once, and
  // the corner table of the NEXT tile is looked up while the matrix pipe works on layers 2 and 3 of the current one.
  int pt0 = 0;
  long long q0 = 0;                        // first global (point, hypothesis) row of the current tile

#ifdef V3D_FUSED_DEBUG
  if (p.dbg_wg && tid == 0) {
    p.dbg_wg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);     // HW_REG_LDS_ALLOC
    p.dbg_wg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    p.dbg_wg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
  }
#endif
  FPHASE_DECL;

  // ---- corner table: 8 hash probes per (row, level), as interp_corners_kernel (sparse.hip) -------------------------------------
  // (the level index is kept wave-uniform everywhere: a per-lane index into the kernel-argument array p.lv[] makes the
  // compiler build a per-lane scratch copy of it)
  // A thread owns one corner of kFProbeRows rows on all three levels.  The lookups are four stages of independent loads --
  // (point, batch) of the rows; the levels' minimum corners; the first slot of every probe sequence, key and value together;
  // resolve (+ the rare longer probe sequence) -- instead of six hash_find() calls in a row, each a chain of four dependent global
  // loads.  Measured per workgroup (wave 0, -DV3D_PHASE_TIMING): 36 k of 128 k cycles when the table was built in front of every
  // tile, whatever the load order -- the chain is latency under a loaded memory system, so the stages of the next tile are spread
  // over the matrix phases of layers 2 and 3, where the gather registers are free.
  constexpr int kFProbeRows = kFRows * 8 / kFThreads;
  int corner = tid & 7;
  // what survives from one tile to the next: feature row (-1 = absent) and weight of this thread's corner of its rows
  int ct_row[3][kFProbeRows];
  float ct_w[3][kFProbeRows];
  // the lookups in flight: declared per tile (below) so that nothing but ct_row / ct_w is carried around the tile loop
  struct Probe {
    bool live[kFProbeRows], ok[3][kFProbeRows];
    float px[kFProbeRows], py[kFProbeRows], pz[kFProbeRows], mn[3][kFProbeRows][3];
    int bb[kFProbeRows], found[3][kFProbeRows];
    unsigned long long key[3][kFProbeRows];
    unsigned slot[3][kFProbeRows];
  };
  auto ct_points = [&](Probe& c, int tile) __attribute__((always_inline)) {
    const int tp0 = tile * npt;
    const long long tq0 = (long long)tp0 * n_hyp;
#pragma unroll
    for (int i = 0; i < kFProbeRows; ++i) {
      const int r = (tid >> 3) + i * (kFThreads / 8);
      c.live[i] = r < rows && tq0 + r < n_q;
      const long long q = c.live[i] ? tq0 + r : tq0;
      c.px[i] = p.pts[(size_t)q * 3 + 0]; c.py[i] = p.pts[(size_t)q * 3 + 1]; c.pz[i] = p.pts[(size_t)q * 3 + 2];
      c.bb[i] = (int)p.pts_batch[tp0 + (c.live[i] ? (int)((unsigned)r / (unsigned)n_hyp) : 0)];      // = q / n_hyp
    }
  };
  auto ct_mins = [&](Probe& c) __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int i = 0; i < kFProbeRows; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) c.mn[l][i][a] = p.lv[l].min_pts[c.bb[i] * 3 + a];
  };
  auto ct_keys = [&](Probe& c) __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const float ts = (float)p.lv[l].ts, res = p.lv[l].res;
#pragma unroll
      for (int i = 0; i < kFProbeRows; ++i) {
        // query coordinate in base-voxel units: ((p - min) / x.res) * x.stride   (refinement.py:34-35)
        const float qx = ((c.px[i] - c.mn[l][i][0]) / res) * ts;
        const float qy = ((c.py[i] - c.mn[l][i][1]) / res) * ts;
        const float qz = ((c.pz[i] - c.mn[l][i][2]) / res) * ts;
        const float c0 = floorf(qx / ts) * ts + ((corner & 1) ? ts : 0.f);
        const float c1 = floorf(qy / ts) * ts + ((corner & 2) ? ts : 0.f);
        const float c2 = floorf(qz / ts) * ts + ((corner & 4) ? ts : 0.f);
        float w = 1.f;
        w *= 1.f - fabsf(qx - c0) / ts;
        w *= 1.f - fabsf(qy - c1) / ts;
        w *= 1.f - fabsf(qz - c2) / ts;
        ct_w[l][i] = w;
        c.ok[l][i] = c.live[i] && c0 >= -v3dhash::kGuard && c1 >= -v3dhash::kGuard && c2 >= -v3dhash::kGuard &&
                     c0 <= 60000.f && c1 <= 60000.f && c2 <= 60000.f;
        // (a coordinate outside the key range is never looked up; clamping keeps the int conversions defined)
        const float lo = -(float)v3dhash::kGuard, hi = 60000.f;
        c.key[l][i] = v3dhash::pack_key(c.bb[i], (int)fminf(fmaxf(c0, lo), hi), (int)fminf(fmaxf(c1, lo), hi),
                                        (int)fminf(fmaxf(c2, lo), hi));
        c.slot[l][i] = V3D_FUSED_ABLATE == 4 ? (unsigned)i : v3dhash::hash_u64(c.key[l][i]) & p.lv[l].table.mask;
        c.found[l][i] = -1;
      }
    }
  };
  // All probe sequences of the thread advance in lock step, two slots per round: a round is ONE memory round trip for the six
  // lookups (in turn they cost the sum of their chain lengths: unsuccessful searches -- absent corners are the common case off the
  // surface -- run to the first empty slot, and a wave waits for its longest chain: 39 k cycles per tile, measured).
  auto ct_probe = [&](Probe& c) __attribute__((always_inline)) {
    unsigned pending = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int i = 0; i < kFProbeRows; ++i) pending |= (c.ok[l][i] && V3D_FUSED_ABLATE != 4 ? 1u : 0u) << (l * kFProbeRows + i);
    for (unsigned round = 0; __any(pending != 0) && round <= 0x40000000u; ++round) {
      unsigned long long ka[3][kFProbeRows], kb[3][kFProbeRows];
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < kFProbeRows; ++i) {
          // (a finished lookup re-reads its last slots: unconditional loads keep the six of them in one batch)
          ka[l][i] = p.lv[l].table.keys[c.slot[l][i]];
          kb[l][i] = p.lv[l].table.keys[(c.slot[l][i] + 1) & p.lv[l].table.mask];
        }
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < kFProbeRows; ++i) {
          const unsigned bit = 1u << (l * kFProbeRows + i);
          if (pending & bit) {
            const unsigned s0 = c.slot[l][i], s1 = (s0 + 1) & p.lv[l].table.mask;
            if (ka[l][i] == c.key[l][i]) { c.found[l][i] = (int)s0; pending &= ~bit; }
            else if (ka[l][i] == v3dhash::kEmpty) pending &= ~bit;
            else if (kb[l][i] == c.key[l][i]) { c.found[l][i] = (int)s1; pending &= ~bit; }
            else if (kb[l][i] == v3dhash::kEmpty) pending &= ~bit;
            else c.slot[l][i] = (s1 + 1) & p.lv[l].table.mask;
          }
        }
    }
  };
  auto ct_values = [&](Probe& c) __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int i = 0; i < kFProbeRows; ++i) {
        if (V3D_FUSED_ABLATE == 4) ct_row[l][i] = c.ok[l][i] ? (int)(c.key[l][i] >> 32) & 1023 : -1;
        else ct_row[l][i] = c.found[l][i] >= 0 ? p.lv[l].table.vals[c.found[l][i]] : -1;
      }
  };
  // an absent corner (or a padding row) reads feature row 0 with weight 0: the gathers below are unconditional
  auto ct_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int i = 0; i < kFProbeRows; ++i) {
        const int r = (tid >> 3) + i * (kFThreads / 8);
        crow[fused_corner_index(r, l, corner)] = ct_row[l][i] < 0 ? 0 : ct_row[l][i];
        cw[fused_corner_index(r, l, corner)] = ct_row[l][i] < 0 ? 0.f : ct_w[l][i];
      }
  };

  // ---- layer 1: K = 3 taps x (C0 + C1 + C2 + c_feat) channels, produced chunk by chunk ----------------------------------
  int srow = tid >> 3, sc4 = (tid & 7) * 4;
  const int cb1 = p.lv[0].C, cb2 = cb1 + p.lv[1].C, cb3 = cb2 + p.lv[2].C;
  // Gather registers: the level chunks (8 corner rows per staging row) and the per-point feature chunks (one row) have their
  // own registers and their own loops below -- with one conditional producer the compiler routed the gathers through
  // temporaries and waited for ALL of them before the chunk's first MFMA (no overlap).
  f32x4 xr[kFStageRows][8], xf[kFStageRows];
  const int nkc_lv = cb3 / 32;             // chunks produced by interpolation; chunks nkc_lv .. nkc1 - 1 copy pts_feat
  // level of a 32-channel chunk: wave-uniform (chunk boundaries are multiples of 32)
  auto level_of = [&](int kc) __attribute__((always_inline)) {
    const int c = kc * 32;
    return c < cb1 ? 0 : c < cb2 ? 1 : 2;
  };
  // part = -1: all gathers of the chunk; 0 / 1 / 2: a third of them (issued between the taps of the previous chunk: one burst of
  // 16 x 1 KB per wave from all eight waves of the CU backs up the CU's single vector-memory path and the wave cannot issue its
  // matrix instructions behind it -- measured 1.6 k cycles for the burst)
  auto issue_lv = [&](int kc, int part) __attribute__((always_inline)) {
    const int l = level_of(kc);
    const float* const feats = l == 0 ? p.lv[0].feats : l == 1 ? p.lv[1].feats : p.lv[2].feats;
    const int C = l == 0 ? p.lv[0].C : l == 1 ? p.lv[1].C : p.lv[2].C;
    const int lc = kc * 32 - (l == 0 ? 0 : l == 1 ? cb1 : cb2) + sc4;
#pragma unroll
    for (int ps = 0; ps < kFStageRows; ++ps) {
      const int r = srow + (kFThreads / 8) * ps;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (part >= 0 && ((ps * 8 + k) * 3) / (kFStageRows * 8) != part) continue;
        const int cr = V3D_FUSED_ABLATE == 2 ? (crow[fused_corner_index(r, l, k)] & 7) : crow[fused_corner_index(r, l, k)];
        xr[ps][k] = *reinterpret_cast<const f32x4*>(feats + (size_t)cr * C + lc);
      }
    }
  };
  auto issue_feat = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int ps = 0; ps < kFStageRows; ++ps) {
      const int r = srow + (kFThreads / 8) * ps;
      xf[ps] = (r < rows && q0 + r < n_q)
                   ? *reinterpret_cast<const f32x4*>(p.pts_feat + (size_t)(q0 + r) * p.c_feat + (kc * 32 - cb3 + sc4))
                   : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // split one staging row's 4 channels and commit them to the staging tile of chunk kc (tile kc & 1)
  auto commit_row = [&](int kc, int r, const float (&v)[4]) __attribute__((always_inline)) {
#ifdef V3D_FUSED_DEBUG
    if (p.dbg_x && r < rows && q0 + r < n_q)
      *reinterpret_cast<f32x4*>(p.dbg_x + (size_t)(q0 + r) * (p.nkc1 * 32) + kc * 32 + sc4) = (f32x4){v[0], v[1], v[2], v[3]};
#endif
    u32x2 hi, lo;
    fused_split4(v, hi, lo);
    const int kg = sc4 >> 3, half = (sc4 >> 2) & 1;
    const int slot = kg ^ (((r >> 3) & 1) * 3);
    u32x2* x2 = reinterpret_cast<u32x2*>(xq + (kc & 1) * kFStageSlots);
    x2[(r * 4 + slot) * 2 + half] = hi;
    x2[((kFRT + r) * 4 + slot) * 2 + half] = lo;
  };
  auto commit_lv = [&](int kc) __attribute__((always_inline)) {
    const int l = level_of(kc);
#pragma unroll
    for (int ps = 0; ps < kFStageRows; ++ps) {
      const int r = srow + (kFThreads / 8) * ps;
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {      // corner order x fastest; absent corners add nothing (no renormalisation)
        const float w = cw[fused_corner_index(r, l, k)];
        a = __builtin_elementwise_fma(xr[ps][k], (f32x4){w, w, w, w}, a);
      }
      const float v[4] = {a.x, a.y, a.z, a.w};
      commit_row(kc, r, v);
    }
  };
  auto commit_feat = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int ps = 0; ps < kFStageRows; ++ps) {
      const float v[4] = {xf[ps].x, xf[ps].y, xf[ps].z, xf[ps].w};
      commit_row(kc, srow + (kFThreads / 8) * ps, v);
    }
  };

  // per column block: does tap 0 / tap 2 of this lane's output row stay inside its hypothesis group?
  unsigned tapmask = 0;
#pragma unroll
  for (int nb = 0; nb < kFNB; ++nb) {
    const int hh = ((cb0 + nb) * 16 + jn) % n_hyp;
    tapmask |= (hh >= 1 ? 1u : 0u) << (2 * nb);
    tapmask |= (hh + 1 < n_hyp ? 1u : 0u) << (2 * nb + 1);
  }
  // Weight fragments: a ring of three tap steps.  The (layer, chunk, tap) steps form one cyclic sequence through all the tiles of
  // the workgroup; the fragments of step g + 2 are requested when step g starts, so an L2 round trip has two steps' worth of
  // matrix instructions (2 x 24 x 16 cycles of this wave alone) to hide behind.  Three taps per chunk = three ring slots: the
  // slot of a tap is a compile-time constant.
  u32x4 a_ring[3][2 * kFMBW];
  auto load_a = [&](u32x4 (&a)[2 * kFMBW], const float* wp, int nkc, int t, int kc) __attribute__((always_inline)) {
    const u32x4* w = reinterpret_cast<const u32x4*>(wp + (V3D_FUSED_ABLATE == 3 ? (size_t)0 : (size_t)(t * nkc + kc) * kFWslab)) + lane;
#pragma unroll
    for (int m = 0; m < kFMBW; ++m) {
      a[m] = w[(wave * kFMBW + m) * 64];
      a[kFMBW + m] = w[(kFMB + wave * kFMBW + m) * 64];
    }
  };
  f32x4 acc[kFNB][kFMBW];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < kFNB; ++nb)
#pragma unroll
      for (int m = 0; m < kFMBW; ++m) acc[nb][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  bool has_next = false;                   // another tile follows the current one (wave-uniform)
  // the fragments of the step two after (layer, kc, t); layer 0 = the first Conv1d (nkc1 chunks), 1 and 2 = the 128 -> 128 ones;
  // behind the last layer the sequence starts over for the next tile
  auto prefetch_a = [&](u32x4 (&a)[2 * kFMBW], int layer, int kc, int t) __attribute__((always_inline)) {
    int t2 = t + 2, kc2 = kc, l2 = layer;
    if (t2 >= 3) { t2 -= 3; ++kc2; }
    if (kc2 >= (layer == 0 ? p.nkc1 : 4)) { kc2 = 0; ++l2; }
    if (l2 == 3) {
      if (!has_next) return;
      l2 = 0;
    }
    load_a(a, l2 == 0 ? p.w[0] : l2 == 1 ? p.w[1] : p.w[2], l2 == 0 ? p.nkc1 : 4, t2, kc2);
  };
  auto mfma_block = [&](const u32x4 (&a_cur)[2 * kFMBW], const bf16x8 b_hi, const bf16x8 b_lo, int nb) __attribute__((always_inline)) {
#if defined(V3D_FUSED_NOMFMA) || V3D_FUSED_ABLATE == 1      // developer experiment: B fragments are read, no matrix instruction is issued
    asm volatile("" : : "v"(b_hi), "v"(b_lo));
    return;
#endif
#pragma unroll
    for (int m = 0; m < kFMBW; ++m) {
      const bf16x8 a_hi = __builtin_bit_cast(bf16x8, a_cur[m]), a_lo = __builtin_bit_cast(bf16x8, a_cur[kFMBW + m]);
      acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc[nb][m], 0, 0, 0);
      acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc[nb][m], 0, 0, 0);
      acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc[nb][m], 0, 0, 0);
    }
  };
  // folded BatchNorm biases and head weights: tile-invariant, parked in LDS once (a global load in front of every activation
  // store / head is an exposed L2 round trip per layer and tile; registers are what this kernel does not have)
  for (int i = tid; i < 3 * kFH; i += kFThreads) cbias[i] = (i < kFH ? p.bias[0] : i < 2 * kFH ? p.bias[1] : p.bias[2])[i % kFH];
  for (int i = tid; i < 3 * kFH; i += kFThreads) chead[i] = p.head_w[i];
  if (tid == 0) chead[3 * kFH] = p.head_b[0];
  if (tid < 8) chead[3 * kFH + 4 + tid] = (p.vals && tid < p.n_hyp) ? p.vals[tid] : 0.f;      // offset values of the expectation
  // bias + ReLU of this wave's 32 channels x 64 rows -> split -> LDS activation buffer (B-fragment order, 16-byte slots of 8
  // channels, slot index XOR-ed with the row so that the 16 row-lanes of a ds_read_b128 hit 16 different slots)
  auto store_act = [&](u32x4* dst, const float* bias) __attribute__((always_inline)) {
    u32x2* d2 = reinterpret_cast<u32x2*>(dst);
#pragma unroll
    for (int nb = 0; nb < kFNB; ++nb) {
      const int r = (cb0 + nb) * 16 + jn;
#pragma unroll
      for (int mw = 0; mw < kFMBW; ++mw) {
        const int co0 = (wave * kFMBW + mw) * 16 + kq * 4;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + co0);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[nb][mw][k] + bq[k], 0.f);
        u32x2 hi, lo;
        fused_split4(v, hi, lo);
        const int slot = (co0 >> 3) ^ (r & 15), half = (co0 >> 2) & 1;
        d2[(r * 16 + slot) * 2 + half] = hi;
        d2[((kFRT + r) * 16 + slot) * 2 + half] = lo;
      }
    }
  };

  // the three taps of chunk kc: MFMAs on its staging tile, A fragments two taps ahead
  auto mfma_chunk = [&](int kc, auto tap_hook) __attribute__((always_inline)) {
    const u32x4* const xs = xq + (kc & 1) * kFStageSlots;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      prefetch_a(a_ring[(t + 2) % 3], 0, kc, t);
      tap_hook(t);
#pragma unroll
      for (int nb = 0; nb < kFNB; ++nb) {
        const bool inside = t == 1 || ((tapmask >> (2 * nb + (t >> 1))) & 1u);
        const int R = inside ? (cb0 + nb) * 16 + jn + t - 1 : kFZero;
        const int slot = R * 4 + (kq ^ (((R >> 3) & 1) * 3));
        mfma_block(a_ring[t], __builtin_bit_cast(bf16x8, xs[slot]), __builtin_bit_cast(bf16x8, xs[kFRT * 4 + slot]), nb);
      }
      FPHASE_FINE(2 + t);
    }
  };
  // a 128 -> 128 layer on the activations in `src`; `hook(kc)` runs in front of every chunk (the next tile's corner-table stages)
  auto dense_layer = [&](int layer, const u32x4* src, auto hook) __attribute__((always_inline)) {
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      hook(kc);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        prefetch_a(a_ring[(t + 2) % 3], layer, kc, t);
#pragma unroll
        for (int nb = 0; nb < kFNB; ++nb) {
          const bool inside = t == 1 || ((tapmask >> (2 * nb + (t >> 1))) & 1u);
          const int R = inside ? (cb0 + nb) * 16 + jn + t - 1 : kFZero;
          const int slot = R * 16 + ((kc * 4 + kq) ^ (R & 15));
          mfma_block(a_ring[t], __builtin_bit_cast(bf16x8, src[slot]), __builtin_bit_cast(bf16x8, src[kFRT * 16 + slot]), nb);
        }
      }
    }
  };

  int hpt = tid >> 5, hl32 = tid & 31, hc0 = hl32 * 4;       // head: 32 lanes per point, 4 channels per lane
  auto refresh_lane_ids = [&]() __attribute__((always_inline)) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    tid = t; lane = t & 63; kq = lane >> 4; jn = lane & 15; corner = t & 7; srow = t >> 3; sc4 = (t & 7) * 4;
    hpt = t >> 5; hl32 = t & 31; hc0 = hl32 * 4;
  };

  // Tile walk: workgroups are dealt round-robin to the 8 XCDs, so XCD x = blockIdx.x % 8 takes the x-th contiguous eighth of the
  // tiles (consecutive tiles are neighbouring pixels of one view): the corner rows its workgroups gather then come from the part
  // of the scene a few views see and stay in that XCD's 4 MB L2, instead of every XCD streaming all three feature tables.
  const int n_xcd = V3D_FUSED_ABLATE != 7 && gridDim.x % 8 == 0 ? 8 : 1;
  const int tiles_per_xcd = (n_tiles + n_xcd - 1) / n_xcd, tile_step = gridDim.x / n_xcd;
  const int tile_end = min(n_tiles, ((int)blockIdx.x % n_xcd + 1) * tiles_per_xcd);
  int tile = ((int)blockIdx.x % n_xcd) * tiles_per_xcd + (int)blockIdx.x / n_xcd;
  if (tile >= tile_end) return;
#ifdef V3D_FUSED_SKEW
  // developer experiment: the second half of the grid starts V3D_FUSED_SKEW x 8 k cycles late, so that the two workgroups of a CU
  // do not walk through the same phases at the same time
  if (blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < V3D_FUSED_SKEW; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  {
    Probe c;
    ct_points(c, tile);
    load_a(a_ring[0], p.w[0], p.nkc1, 0, 0);
    load_a(a_ring[1], p.w[0], p.nkc1, 1, 0);
    ct_mins(c);
    ct_keys(c);
    ct_probe(c);
    ct_values(c);
  }
#pragma unroll 1
  for (; tile < tile_end; tile += tile_step) {
    refresh_lane_ids();
    pt0 = tile * npt;
    q0 = (long long)pt0 * n_hyp;
    has_next = tile + tile_step < tile_end;
    // every wave is past the barrier behind layer 3 of the previous tile: the second buffer (staging tiles, corner table) is free
    if (tid < 2 * 2 * 4) xq[(tid >> 3) * kFStageSlots + (((tid >> 2) & 1) * kFRT + kFZero) * 4 + (tid & 3)] = (u32x4){0u, 0u, 0u, 0u};
    ct_store();
    zero_acc();
    __syncthreads();                         // corner table ready
    FPHASE_MARK(0);
    issue_lv(0, -1);
    commit_lv(0);
    __syncthreads();
    FPHASE_MARK(1);
    // chunk kc: its staging tile is complete; the next chunk's gathers fly during this chunk's MFMAs and are committed to the
    // OTHER staging tile behind them (last read by chunk kc - 1, which every wave finished before the barrier in front of chunk
    // kc): one barrier per chunk
#pragma unroll 1
    for (int kc = 0; kc < nkc_lv; ++kc) {
#if V3D_FUSED_ABLATE == 8
      if (kc + 1 < nkc_lv) issue_lv(kc + 1, -1);
      else if (nkc_lv < p.nkc1) issue_feat(nkc_lv);
      FPHASE_FINE(1);
      mfma_chunk(kc, [&](int) __attribute__((always_inline)) {});
#else
      if (kc + 1 >= nkc_lv && nkc_lv < p.nkc1) issue_feat(nkc_lv);
      FPHASE_FINE(1);
      mfma_chunk(kc, [&](int t) __attribute__((always_inline)) {
        if (kc + 1 < nkc_lv) issue_lv(kc + 1, t);
      });
#endif
      FPHASE_MARK(2);
      if (kc + 1 < nkc_lv) commit_lv(kc + 1);
      else if (nkc_lv < p.nkc1) commit_feat(nkc_lv);
      FPHASE_FINE(5);
      __syncthreads();
      FPHASE_FINE(6);
      FPHASE_MARK(1);
    }
#pragma unroll 1
    for (int kc = nkc_lv; kc < p.nkc1; ++kc) {
      if (kc + 1 < p.nkc1) issue_feat(kc + 1);
      FPHASE_FINE(1);
      mfma_chunk(kc, [&](int) __attribute__((always_inline)) {});
      FPHASE_MARK(2);
      if (kc + 1 < p.nkc1) commit_feat(kc + 1);
      FPHASE_FINE(5);
      __syncthreads();
      FPHASE_FINE(6);
      FPHASE_MARK(1);
    }
    // (the fp32 output of the previous tile's last layer covered the zero row of the first buffer)
    if (tid < 2 * 16) actA[((tid >> 4) * kFRT + kFZero) * 16 + (tid & 15)] = (u32x4){0u, 0u, 0u, 0u};
    store_act(actA, cbias);
    if (tid < 2 * 16) actB[((tid >> 4) * kFRT + kFZero) * 16 + (tid & 15)] = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();        // act1 complete; staging tiles / corner table (aliasing the second buffer) no longer needed
    FPHASE_MARK(3);

    // ---- layers 2 and 3: K = 3 taps x 128 channels read from LDS; the next tile's corner table is looked up on the side ------------
    const int next_tile = tile + tile_step;
    Probe c;
    zero_acc();
    dense_layer(1, actA, [&](int kc) __attribute__((always_inline)) {
      if (has_next && kc == 0) ct_points(c, next_tile);
      if (has_next && kc == 2) ct_mins(c);
    });
    FPHASE_MARK(4);
    if (V3D_FUSED_ABLATE != 5) {
      store_act(actB, cbias + kFH);
      __syncthreads();
      FPHASE_MARK(3);
      zero_acc();
      dense_layer(2, actB, [&](int kc) __attribute__((always_inline)) {
        if (has_next && kc == 0) ct_keys(c);
        if (has_next && kc == 1) ct_probe(c);
        if (has_next && kc == 3) ct_values(c);
      });
      FPHASE_MARK(5);
    }
    {
      // last layer: fp32 [64 rows][128] into the first buffer (act1 is dead: every wave passed the barrier after layer 2)
      float* const of = reinterpret_cast<float*>(actA);
#pragma unroll
      for (int nb = 0; nb < kFNB; ++nb)
#pragma unroll
        for (int mw = 0; mw < kFMBW; ++mw) {
          const int co0 = (wave * kFMBW + mw) * 16 + kq * 4;
          const f32x4 bq = *reinterpret_cast<const f32x4*>(cbias + 2 * kFH + co0);
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[nb][mw][k] + bq[k], 0.f);
          *reinterpret_cast<f32x4*>(of + ((cb0 + nb) * 16 + jn) * kFH + co0) = v;
        }
    }
    __syncthreads();
    FPHASE_MARK(3);

    // ---- head: Conv1d(128 -> 1, k3, pad 1, bias) over the hypotheses, softmax, expectation (32 lanes per point) ----------------
    {
      const float* const of = reinterpret_cast<const float*>(actA);
      float score[8], hw0[4], hw1[4], hw2[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { hw0[k] = chead[(hc0 + k) * 3]; hw1[k] = chead[(hc0 + k) * 3 + 1]; hw2[k] = chead[(hc0 + k) * 3 + 2]; }
      const float head_b = chead[3 * kFH];
      // (32 lanes per point, kFThreads / 32 points per pass: a tile of 9 points takes a second pass for its last one)
#pragma unroll 1
      for (int hp = hpt; hp < npt; hp += kFThreads / 32) {
#pragma unroll
      for (int h = 0; h < 8; ++h) score[h] = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        if (h < n_hyp) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(of + (hp * n_hyp + h) * kFH + hc0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // out[h'] = sum_t in[h' + t - 1] w[t]  =>  in[h] feeds out[h+1] (t=0), out[h] (t=1), out[h-1] (t=2)
            if (h + 1 < 8) score[h + 1] += x[k] * hw0[k];
            score[h] += x[k] * hw1[k];
            if (h > 0) score[h - 1] += x[k] * hw2[k];
          }
        }
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float v = score[h];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        score[h] = v + head_b;
      }
      if (hl32 == 0 && pt0 + hp < p.n_pts) {
        float m = -INFINITY;
#pragma unroll
        for (int h = 0; h < 8; ++h) if (h < n_hyp) m = fmaxf(m, score[h]);
        float ex[8], sum = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) {                 // every exponential once; same values, same summation order as before
          ex[h] = h < n_hyp ? expf(score[h] - m) : 0.f;
          if (h < n_hyp) sum += ex[h];
        }
        float e = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          if (h < n_hyp) {
            const float pr = ex[h] / sum;
            p.preds[(size_t)(pt0 + hp) * n_hyp + h] = pr;
            e += chead[3 * kFH + 4 + h] * pr;         // (offset values parked in LDS: a global load per hypothesis sat in this chain)
          }
        }
        if (p.expect) p.expect[pt0 + hp] = e;
      }
      }
    }
    FPHASE_MARK(6);
    // the next tile's first writes into the second buffer are safe (every wave passed the barrier behind layer 3); its writes into
    // the first buffer (activation stores of layer 1) come several barriers after this tile's head reads
  }
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

#ifdef V3D_FUSED_DEBUG
static float* g_fused_dbg_x = nullptr;
static unsigned* g_fused_dbg_wg = nullptr;
extern "C" void v3d_debug_fused_dump(float* x) { g_fused_dbg_x = x; }
extern "C" void v3d_debug_fused_dump_wg(unsigned* x) { g_fused_dbg_wg = x; }
#endif

extern "C" int v3d_decoder_fused_f32(const v3d_gemm_weights* const* layers_host, const float* head_weight,
                                     const float* head_bias, const void* const* level_table_host,
                                     const int* level_n_host, const float* const* level_feats_host,
                                     const int* level_C_host, const int* level_stride_host,
                                     const float* const* level_min_pts_host, const float* level_res_host,
                                     const float* pts, const int64_t* pts_batch, const float* pts_feat, int c_feat,
                                     int n_pts, int n_hyp, const float* offset_vals, float* preds, float* expect,
                                     void* stream) {
  V3D_REQUIRE(layers_host && head_weight && head_bias && level_table_host && level_n_host && level_feats_host &&
                  level_C_host && level_stride_host && level_min_pts_host && level_res_host && pts && pts_batch && preds,
              V3D_ERR_BAD_ARG, "v3d_decoder_fused_f32: null argument");
  V3D_REQUIRE(n_pts >= 0 && n_hyp >= 1 && n_hyp <= 8, V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: n_hyp=%d (1..8)", n_hyp);
  V3D_REQUIRE(c_feat >= 0 && c_feat % 32 == 0 && (c_feat == 0 || pts_feat), V3D_ERR_UNSUPPORTED,
              "v3d_decoder_fused_f32: c_feat=%d must be a multiple of 32 (with pts_feat given)", c_feat);
  V3D_REQUIRE(!expect || offset_vals, V3D_ERR_BAD_ARG, "v3d_decoder_fused_f32: expect without offset_vals");
  FusedParams p;
  memset(&p, 0, sizeof(p));
  int k1 = c_feat;
  for (int l = 0; l < 3; ++l) {
    V3D_REQUIRE(level_table_host[l] && level_feats_host[l] && level_min_pts_host[l] && level_n_host[l] > 0 &&
                    level_C_host[l] > 0 && level_C_host[l] % 32 == 0 && level_stride_host[l] > 0 && level_res_host[l] > 0.f,
                V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: level %d (channels must be a multiple of 32)", l);
    p.lv[l].table = v3dhash::table_view(const_cast<void*>(level_table_host[l]), level_n_host[l]);
    p.lv[l].feats = level_feats_host[l]; p.lv[l].min_pts = level_min_pts_host[l]; p.lv[l].res = level_res_host[l];
    p.lv[l].C = level_C_host[l]; p.lv[l].ts = level_stride_host[l];
    k1 += level_C_host[l];
  }
  for (int l = 0; l < 3; ++l) {
    const v3d_gemm_weights* h = layers_host[l];
    V3D_REQUIRE(h && h->n_seg == 3 && h->N == kFH && h->MBW == kFMBW && h->has_bias && h->K == (l == 0 ? k1 : kFH) &&
                    h->KP == h->K,
                V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: layer %d must be a packed Conv1d(k3) %d -> 128 with bias", l,
                l == 0 ? k1 : kFH);
    p.w[l] = h->dev + h->bf_ofs;
    p.bias[l] = h->dev + h->bias_ofs;
  }
  p.nkc1 = k1 / 32;
  p.pts = pts; p.pts_batch = (const long long*)pts_batch; p.pts_feat = pts_feat; p.c_feat = c_feat;
  p.n_pts = n_pts; p.n_hyp = n_hyp;
  p.head_w = head_weight; p.head_b = head_bias; p.vals = offset_vals; p.preds = preds; p.expect = expect;
#ifdef V3D_FUSED_DEBUG
  p.dbg_x = g_fused_dbg_x;
  p.dbg_wg = g_fused_dbg_wg;
#endif
  if (n_pts == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  // developer switch (scripts/micro/fused_decoder_stress.py): dynamic LDS request in KB
  static const size_t lds_bytes = getenv("V3D_FUSED_LDS_KB") ? (size_t)atoi(getenv("V3D_FUSED_LDS_KB")) * 1024 : kFLdsBytes;
  V3D_REQUIRE(lds_bytes >= kFUsedLdsBytes && lds_bytes <= 160 * 1024, V3D_ERR_BAD_ARG, "V3D_FUSED_LDS_KB out of range");
  // per device: the dynamic-LDS opt-in and the number of workgroups the device holds at once (160 KB / lds_bytes per CU)
  static int resident_of[64] = {0};
  int dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "v3d_decoder_fused_f32: device ordinal %d", dev);
  if (!resident_of[dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)decoder_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_bytes));
    int n_cu = 0;
    V3D_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int per_cu = getenv("V3D_FUSED_WG_PER_CU") ? atoi(getenv("V3D_FUSED_WG_PER_CU")) : (int)(160 * 1024 / lds_bytes);
    resident_of[dev] = (n_cu > 0 ? n_cu : 256) * (per_cu > 0 ? per_cu : 1);
  }
  const int resident = resident_of[dev];
  {
    // persistent tile walk: workgroup b takes tiles b, b + grid, ... (equal work per tile, so a static split is balanced)
    const int npt = kFRows / n_hyp;                    // as in the kernel
    const int n_tiles = (n_pts + npt - 1) / npt;
    v3d::TimedScope ts("decoder_fused", s);
    decoder_fused_kernel<<<n_tiles < resident ? n_tiles : resident, kFThreads, lds_bytes, s>>>(p);
  }
  V3D_CHECK_LAUNCH("decoder_fused_kernel");
  return V3D_OK;
}

