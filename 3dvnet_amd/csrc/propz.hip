// PropagationNet as ONE row-marching kernel (round 6) -- SURVEY.md 8f rank 2, stage 3 of the scene driver:
// mv3d/subnetworks/upsampling.py:14-36 (four 3x3 conv + BN + ReLU layers in -> 32 -> 32 -> 32 -> 9, softmax over the 9
// logits, convex combination of the replicate-padded 3x3 depth neighbourhood), applied after a nearest-neighbour resize of the
// depth (mv3d/eval-3dvnet.py:101-125).
//
// The per-layer path (costreg.hip: prop_encode_kernel + 4 x convg_bf16x2_kernel<FLAT> + prop_finish_kernel) writes and re-reads
// a 32-channel activation tensor per layer: 0.67 GB each at 256 x 320 x 64 views, 6.2 GB per scene, and spends a quarter of its
// matrix instructions on a zero x tap (K = 4 x taps x 8 channels).  Here a workgroup owns a 40-column strip of one image and
// marches down its rows; the activations of the four layers live in LDS as rings of four rows and never reach HBM:
//
//   waves 0..3 = layers 1..4 (layer 1 shared with wave 7), each with its split-bf16 weight fragments in REGISTERS for the whole kernel
//                (K step = one tap x all 32 input channels: no zero tap; layer 1 flattens (tap, channel) into K);
//                layer l computes row t - 2 l at step t from rows r - 1 .. r + 1 of the ring below it and writes bias + ReLU +
//                hi / lo split into its own ring; the layer-4 wave also does the softmax + 3 x 3 propagation and stores the row;
//   waves 4..6 = helpers: load the guide features + depth of row t + 1 (the nearest-neighbour resize of the depth is two index
//                tables in the addressing), split them and commit row t to the input ring.
//   One barrier per step; H + 8 steps per strip.
//
// Columns: a strip computes 48 columns (3 MFMA column blocks of 16) for 40 outputs -- every layer loses one column per side.
// A ring row is stored as planes of 16-byte chunks (kPLS below: conflict-free B-fragment reads).  Zero padding of every convolution: image-border columns / rows are written as zeros by the producing
// wave (not as "the convolution evaluated outside the image").
#include <cstring>
#include <mutex>

#include "v3d_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTWO = 40;                 // output columns per strip
constexpr int kNBLK = 3, kTC = 16 * kNBLK;   // computed columns: c = 0 .. 47 <-> x = x0 - 4 + c
constexpr int kPSL = kTC + 2;            // pixel slots per ring row (one guard slot on either side)
// LDS layout of a ring row (round 6, second version): PLANES of 16-byte chunks, [chunk][pixel slot] with the pixel slots of a chunk
// 16 bytes apart and the planes kPLS = 1 024 bytes apart.  A split-bf16 row of 32 channels has 8 chunks (hi of channels 8 j .. 8 j + 7
// = chunk j, lo = chunk 4 + j), an fp32 row 8 chunks (channels 4 j .. 4 j + 3).  Why: ds_read_b128 is serviced in four groups of 16
// lanes that MIX two kq values -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) -- so the lanes of a group
// must cover the 64 banks whatever their kq: with a chunk's 16 pixels contiguous (256 bytes = one bank row) and the planes a multiple
// of 256 bytes apart they do, at any x shift.  The first version (144-byte pixel slots: hi + lo + 16 of padding, conflict-free only
// for 16 lanes of ONE kq) lost 46 % of its LDS cycles to 2-way conflicts (`SQ_LDS_BANK_CONFLICT` 1.06e8 of 2.30e8 per launch).
constexpr int kPLS = 1024;               // plane stride (kPSL * 16 = 800 bytes used)
constexpr int kRowB = 8 * kPLS;          // a 32-channel row: 8 planes
constexpr int kLag = 2;                  // rows a layer trails the one below it
constexpr int kDR = 16;                  // rows of the fp32 depth ring
constexpr int kThreads = 512;

// G = padded input channels of layer 1 / 8 (1: image + depth, 3: 16-channel features + depth, 5: 32-channel features + depth)
template <int G>
struct PZ {
  static constexpr int CIN1P = 8 * G;
  static constexpr int R0 = 2 * G * kPLS;                 // the input ring's row: G hi + G lo chunks (fp32: 2 G chunks of 4 channels)
  static constexpr int KS1 = (9 * G + 3) / 4;             // K steps of layer 1: 9 taps x G channel groups, four per step
  static constexpr int RING0 = 0;
  static constexpr int RING1 = 4 * R0;
  static constexpr int RING2 = RING1 + 4 * kRowB;
  static constexpr int RING3 = RING2 + 4 * kRowB;
  static constexpr int DEPTH = RING3 + 4 * kRowB;          // [kDR][kPSL] floats
  static constexpr int LOGIT = DEPTH + kDR * kPSL * 4;     // [kTC][12] floats, private to the layer-4 wave
  static constexpr int LDS = LOGIT + kTC * 12 * 4;
  static_assert(kPSL * 16 <= kPLS && kPLS % 256 == 0, "a plane holds a row's pixel slots; planes are whole bank rows apart");
  static_assert(LDS <= 160 * 1024, "one workgroup per CU");
};

struct PropzParams {
  const float* feat;     // [B, Cf, H, W]
  const float* depth;    // [B, h0, w0] (the depth BEFORE the nearest resize; h0 == H, w0 == W and null tables: already resized)
  const int* iy;         // [H] source row of every output row (null: identity)
  const int* ix;         // [W] source column (null: identity)
  float* out;            // [B, H, W]
  const u32x4* w[4];     // fragment images [K step][16-row block][hi, lo][64 lanes]
  const float* bias[4];  // folded BatchNorm bias, padded to the block count
  int B, Cf, H, W, h0, w0, nstrip;
};

__device__ __forceinline__ unsigned pz_pack(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
// x = hi + lo: hi = RNE_bf16(x), lo = RNE_bf16(x - hi) (the operand split of every split-bf16 kernel of the library)
__device__ __forceinline__ void pz_split4(float a, float b, float c, float d, u32x2& hp, u32x2& lp) {
  hp = (u32x2){pz_pack(a, b), pz_pack(c, d)};
  lp = (u32x2){pz_pack(a - __uint_as_float(hp.x << 16), b - __uint_as_float(hp.x & 0xffff0000u)),
               pz_pack(c - __uint_as_float(hp.y << 16), d - __uint_as_float(hp.y & 0xffff0000u))};
}

// ---- matrix waves -------------------------------------------------------------------------------------------------------------
// F32 (V3D_PRECISION_FP32, the reference's arithmetic type): the same kernel on v_mfma_f32_16x16x4_f32 -- a pixel slot holds its
// channels as fp32 (4 bytes per channel where the split layout has 2 + 2: the same number of 16-byte chunks), the two 16-byte reads of a
// lane are channels 8 q .. 8 q + 3 and 8 q + 4 .. 8 q + 7, the two fragment registers of a (K step, 16-row block) hold the
// weights of k slices 0..3 and 4..7 (slice s multiplies channel 8 kq + s): eight exact-fp32 matrix instructions per (block, co
// block, K step) where the split path has three bf16 ones -- same registers, same LDS, 5.3x the matrix time.
// CB0 / NCB: the 16-row blocks of output channels this wave computes (see propz_kernel for who computes what).
// Round-6 measurements behind the layout, 64 views, ms per net (64x80 cin 33 | 128x160 cin 33 | 256x320 cin 4):
//   one wave per layer, layer 1's 40-channel fragments streamed from L2 through a ring of four K steps   0.239 | 0.475 | 0.967
//   ... its lo fragments in 24 KB of LDS instead                                                         0.207 | 0.410 | 0.963
//   layer 1 split by channel halves over waves 0 and 7, three helper waves, coalesced helper loads       0.163 | 0.345 | 1.04
//   1 / 2 / 4 input rows of prefetch: no difference (the pipeline does not wait for the loads)
//   two of the three split products (timing only): -17 % instead of -33 %: half of a row's time is the critical waves' LDS reads,
//     epilogue and barrier, not matrix instructions
//   EIGHT matrix waves (layers 2 / 3 by channel halves, layers 1 / 4 in column-block pieces on the loader waves)
//     0.215 | 0.431 | 1.41: every B fragment is then read from LDS by two waves and the kernel becomes LDS-bound
//   ring rows as planes of 16-byte chunks (kPLS): SQ_LDS_BANK_CONFLICT 1.06e8 -> 2.3e7 per launch, time unchanged           0.163 | 0.343 | 0.98
//   six matrix waves balanced by COLUMN BLOCKS (no fragment read twice; 135 instead of 162 matrix instructions on the busiest
//     SIMD, loaders merged into the light waves)                                                         0.200 | 0.400 | 1.10
//     (exact fp32, where the matrix instructions are 5x longer: 3.14 instead of 3.26 ms) -- with split operands a row is a
//     serial chain per wave (barrier, first LDS reads, 162 matrix instructions, epilogue, barrier): more work per loader wave
//     lengthens the slowest chain; what would shorten it is two rows per step, which the LDS does not hold
//   two passes over K (column blocks {0, 1}, then {2} with the first pass's epilogues between its K steps): +-0
//   the step barrier as `s_waitcnt lgkmcnt(0); s_barrier` instead of __syncthreads() (no wait for the loaders' requests and
//     layer 4's stores): +-0.  `SQ_VALU_MFMA_BUSY_CYCLES` says 44 % of all SIMD cycles = 62 % on the two SIMDs of layers 2 / 3
//   ablations (-DV3D_PZ_ABLATE, timing only): two of three products 0.143 | 0.291 | 0.890; NO matrix instructions 0.099 | 0.186 |
//     0.481 (then a row takes one global-load latency: the loaders' requests); no exp / division in the softmax: +-0.  A row of
//     the full kernel = the layer-2 / layer-3 wave's chain: barrier -> first fragment reads -> 162 matrix instructions x 16
//     cycles -> epilogue -> barrier, 1.8 us of which the matrix instructions are ~1.3
template <int G, int LAYER, bool F32, int CB0, int NCB>
__device__ __forceinline__ void pz_matrix_role(const PropzParams& p, unsigned char* smem, int lane, int item0, int item_step, int n_items) {
  typedef PZ<G> Z;
  constexpr bool L1 = LAYER == 1;
  constexpr int KS = L1 ? Z::KS1 : 9, NCBL = LAYER == 4 ? 1 : 2;      // K steps; 16-row blocks of the whole layer
  static_assert(CB0 + NCB <= NCBL, "co blocks");
  constexpr int RBI = L1 ? Z::R0 : kRowB;                     // row bytes of the ring this layer reads
  constexpr int CGB = F32 ? 2 * kPLS : kPLS;                  // bytes between the first reads of consecutive 8-channel groups
  constexpr int LO = F32 ? kPLS : (L1 ? G : 4) * kPLS;        // first -> second 16-byte read of a lane (hi -> lo planes / channels +4)
  constexpr int RIN = LAYER == 1 ? Z::RING0 : LAYER == 2 ? Z::RING1 : LAYER == 3 ? Z::RING2 : Z::RING3;
  constexpr int ROUT = LAYER == 1 ? Z::RING1 : LAYER == 2 ? Z::RING2 : Z::RING3;
  const int kq = lane >> 4, jn = lane & 15;

  // the wave's weight fragments: registers, once
  u32x4 a[KS][NCB][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int part = 0; part < 2; ++part) a[ks][cb][part] = p.w[LAYER - 1][((ks * NCBL + CB0 + cb) * 2 + part) * 64 + lane];
  float bias[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[cb][r] = p.bias[LAYER - 1][(CB0 + cb) * 16 + 4 * kq + r];

  // B-fragment addressing.  Layers 2..4: K step = tap (ky, kx), the lane's 8 k values = channels 8 kq .. 8 kq + 7 of pixel
  // (column jn + kx - 1): slot index 1 + blk * 16 + jn + kx - 1.  Layer 1: k = tap * CIN1P + channel, the lane's 8 k values =
  // channel group (ks * 4 + kq) % G of tap (ks * 4 + kq) / G (a tap >= 9 carries zero weights and reads tap 0's finite data).
  unsigned lofs[L1 ? KS : 1];
  unsigned lky = 0u;                                            // layer 1: the lane's tap row ky of K step ks in bits 2 ks, 2 ks + 1
  if constexpr (L1) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k8 = ks * 4 + kq;
      const int tap = k8 / G < 9 ? k8 / G : 0, cg = k8 / G < 9 ? k8 % G : 0;
      lofs[ks] = (unsigned)((jn + tap % 3) * 16 + cg * CGB);
      lky |= (unsigned)(tap / 3) << (2 * ks);
    }
  } else {
    lofs[0] = (unsigned)(jn * 16 + kq * CGB);
  }
  // this lane's 4 output channels (16 cb + 4 kq ..) inside block 0, co block 0: fp32 chunk kq; split: half kq & 1 of hi chunk kq >> 1
  const unsigned wofs = (unsigned)((1 + jn) * 16 + (F32 ? kq * kPLS : (kq >> 1) * kPLS + (kq & 1) * 8));

  for (int item = item0; item < n_items; item += item_step) {
    const int b = item / p.nstrip, x0 = (item - b * p.nstrip) * kTWO;
    __syncthreads();                                            // the rings are zeroed (helpers + everyone, see the kernel body)
    float colmask[kNBLK];                                       // 1 inside the image, 0 outside (zero padding of the next layer)
#pragma unroll
    for (int blk = 0; blk < kNBLK; ++blk) {
      const int x = x0 - 4 + blk * 16 + jn;
      colmask[blk] = (x >= 0 && x < p.W) ? 1.f : 0.f;
    }
    for (int t = 0; t < p.H + 4 * kLag; ++t) {                  // layer 4 finishes row H - 1 at step H - 1 + 4 kLag
      const int r = t - kLag * LAYER;
      if (r >= 0 && r < p.H) {
        unsigned rb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rb[ky] = (unsigned)(RIN + ((r + ky - 1) & 3) * RBI);
        f32x4 acc[kNBLK][NCB];
#pragma unroll
        for (int blk = 0; blk < kNBLK; ++blk)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[blk][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 bh[kNBLK], bl[kNBLK];
          unsigned base;
          if constexpr (L1) {
            const unsigned ky = (lky >> (2 * ks)) & 3u;
            base = (ky == 0u ? rb[0] : ky == 1u ? rb[1] : rb[2]) + lofs[ks];
          }
          else base = rb[ks / 3] + lofs[0] + (unsigned)((ks % 3) * 16);
#pragma unroll
          for (int blk = 0; blk < kNBLK; ++blk) {
            bh[blk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + base + blk * 256));
            bl[blk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + base + blk * 256 + LO));
          }
          u32x4 alo[NCB];
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) alo[cb] = a[ks][cb][1];
          if constexpr (F32) {
            // eight k slices per (block, co block), round robin over the accumulators
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
#pragma unroll
              for (int blk = 0; blk < kNBLK; ++blk)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                  const float av = __uint_as_float(sl < 4 ? a[ks][cb][0][sl & 3] : alo[cb][sl & 3]);
                  const float bv = __uint_as_float(sl < 4 ? __builtin_bit_cast(u32x4, bh[blk])[sl & 3] : __builtin_bit_cast(u32x4, bl[blk])[sl & 3]);
                  acc[blk][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[blk][cb], 0, 0, 0);
                }
          } else {
          // three products per (block, co block), round robin over the accumulators (no back-to-back dependent pair)
#ifndef V3D_PZ_ABLATE
#define V3D_PZ_ABLATE 0      // developer ablations (timing only): 1 = two of the three split products, 2 = none (everything but the matrix instructions), 3 = none and no exp / division in the softmax, 4 = all products but that softmax
#endif
#pragma unroll
          for (int prod = 0; prod < ((V3D_PZ_ABLATE == 2 || V3D_PZ_ABLATE == 3) ? 0 : V3D_PZ_ABLATE == 1 ? 2 : 3); ++prod)
#pragma unroll
            for (int blk = 0; blk < kNBLK; ++blk)
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, prod == 2 ? alo[cb] : a[ks][cb][0]);
                acc[blk][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, prod == 1 ? bl[blk] : bh[blk], acc[blk][cb], 0, 0, 0);
              }
          }
        }
        if constexpr (LAYER < 4) {
          // bias + ReLU, zero outside the image, hi / lo split -> this layer's ring: 4 channels = 8 bytes of hi, 8 of lo
          unsigned char* const orow = smem + ROUT + (r & 3) * kRowB + wofs;
#pragma unroll
          for (int blk = 0; blk < kNBLK; ++blk)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
              float v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[blk][cb][q] + bias[cb][q], 0.f) * colmask[blk];
              if constexpr (F32) {
                *reinterpret_cast<f32x4*>(orow + blk * 256 + (CB0 + cb) * 4 * kPLS) = (f32x4){v[0], v[1], v[2], v[3]};
              } else {
                u32x2 hp, lp;
                pz_split4(v[0], v[1], v[2], v[3], hp, lp);
                *reinterpret_cast<u32x2*>(orow + blk * 256 + (CB0 + cb) * 2 * kPLS) = hp;
                *reinterpret_cast<u32x2*>(orow + blk * 256 + (CB0 + cb) * 2 * kPLS + 4 * kPLS) = lp;
              }
            }
        } else {
          // logits (ReLU'd, upsampling.py:6-11,21) -> the wave's scratch [column][12]; then one lane per column: softmax over 9
          // and the weighted sum of the replicate-padded 3 x 3 depth neighbourhood in unfold order (:27-36)
          float* const lg = reinterpret_cast<float*>(smem + Z::LOGIT);
#pragma unroll
          for (int blk = 0; blk < kNBLK; ++blk) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[blk][0][q] + bias[0][q], 0.f);
            if (kq < 3) *reinterpret_cast<f32x4*>(lg + (blk * 16 + jn) * 12 + 4 * kq) = v;
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the wave's own LDS writes have landed
          __builtin_amdgcn_wave_barrier();
          const int c = lane;                                   // columns 4 .. 43 are the strip's outputs
          const int x = x0 - 4 + c;
          if (c >= 4 && c < 4 + kTWO && x < p.W) {
            float e[9], m = -3.4e38f;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(lg + c * 12), q1 = *reinterpret_cast<const f32x4*>(lg + c * 12 + 4);
            e[0] = q0[0]; e[1] = q0[1]; e[2] = q0[2]; e[3] = q0[3]; e[4] = q1[0]; e[5] = q1[1]; e[6] = q1[2]; e[7] = q1[3];
            e[8] = lg[c * 12 + 8];
#pragma unroll
            for (int k = 0; k < 9; ++k) m = fmaxf(m, e[k]);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) { e[k] = (V3D_PZ_ABLATE >= 3 ? e[k] - m : expf(e[k] - m)); sum += e[k]; }
            const float* const dr = reinterpret_cast<const float*>(smem + Z::DEPTH);
            float o = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              const int yy = min(max(r + k / 3 - 1, 0), p.H - 1);
              // (the depth ring's columns are already clamped to the image: slot 1 + c + dx holds x + dx clamped)
              o += (V3D_PZ_ABLATE >= 3 ? e[k] * sum : e[k] / sum) * dr[(yy & (kDR - 1)) * kPSL + 1 + c + (k % 3 - 1)];
            }
            p.out[((size_t)b * p.H + r) * p.W + x] = o;
          }
          __builtin_amdgcn_wave_barrier();                      // the scratch is free for the next row
        }
      } else if (LAYER < 4 && CB0 == 0 && r == p.H) {
        // the row below the image: zeros (the next layer's zero padding); the slot still holds row H - 4
        unsigned char* const orow = smem + ROUT + (r & 3) * kRowB;
        for (int i = lane; i < kRowB / 16; i += 64) reinterpret_cast<u32x4*>(orow)[i] = (u32x4){0u, 0u, 0u, 0u};
      }
      __syncthreads();
    }
  }
}

// ---- helper waves: the input ring + the fp32 depth ring ---------------------------------------------------------------------------
template <int G, bool F32, int HT>
__device__ __forceinline__ void pz_helper_role(const PropzParams& p, unsigned char* smem, int htid, int item0, int item_step, int n_items) {
  typedef PZ<G> Z;
  constexpr int NQ = Z::CIN1P / 4;                              // channel quads per pixel
  constexpr int NTASK = kPSL * NQ, NT = (NTASK + HT - 1) / HT;  // (pixel slot, quad) tasks per row, per helper thread
  const size_t plane = (size_t)p.H * p.W;
  for (int item = item0; item < n_items; item += item_step) {
    const int b = item / p.nstrip, x0 = (item - b * p.nstrip) * kTWO;
    // everything the workgroup's LDS holds is zero at the start of a strip (rows -1 of every ring, guard slots)
    for (int i = htid; i < Z::LDS / 16; i += HT) reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
    const float* const fimg = p.feat + (size_t)b * p.Cf * plane;
    const float* const dimg = p.depth + (size_t)b * p.h0 * p.w0;
    // per task: slot, quad, the source column of the features (clamped: replicate for the depth ring, masked for the conv input)
    // kPF rows in flight; task -> (pixel slot, quad) with the slot fastest (the lanes of a load walk along x inside one channel
    // plane).  Same-box A/B: both help the 33-channel nets (0.208 -> 0.163, 0.412 -> 0.343 ms with the layer-1 split) and cost the
    // 4-channel net 11 % (0.99 -> 1.10 ms: half of its "quads" are padding and load nothing) -- so they depend on G.
    constexpr int kPF = G > 1 ? 2 : 1;
    constexpr bool kSlotFast = G > 1;
    float valr[kPF][NT][4];
    float dvalr[kPF];
    auto issue = [&](int y, float (&val)[NT][4], float& dval) __attribute__((always_inline)) {
      const bool row_ok = y >= 0 && y < p.H;
      const int yc = min(max(y, 0), p.H - 1);
      const int ys = p.iy ? p.iy[yc] : yc;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int task = htid + HT * i;
        const int q = kSlotFast ? task / kPSL : task % NQ, ps = kSlotFast ? task - q * kPSL : task / NQ;
        const int x = x0 - 5 + ps;
        const bool ok = row_ok && task < NTASK && x >= 0 && x < p.W;
        const int xc = min(max(x, 0), p.W - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * q + e;
          float v = 0.f;
          if (c < p.Cf) v = fimg[(size_t)c * plane + (size_t)yc * p.W + xc];
          else if (c == p.Cf) v = dimg[(size_t)ys * p.w0 + (p.ix ? p.ix[xc] : xc)];
          val[i][e] = ok ? v : 0.f;
        }
      }
      // the depth ring: one thread per pixel slot, columns clamped to the image (replicate padding, upsampling.py:29)
      if (htid < kPSL) {
        const int xc = min(max(x0 - 5 + htid, 0), p.W - 1);
        dval = dimg[(size_t)ys * p.w0 + (p.ix ? p.ix[xc] : xc)];
      }
    };
    auto commit = [&](int y, const float (&val)[NT][4], float dval) __attribute__((always_inline)) {
      unsigned char* const row = smem + Z::RING0 + (y & 3) * Z::R0;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int task = htid + HT * i;
        if (task < NTASK) {
          const int q = kSlotFast ? task / kPSL : task % NQ, ps = kSlotFast ? task - q * kPSL : task / NQ;
          if constexpr (F32) {
            *reinterpret_cast<f32x4*>(row + q * kPLS + ps * 16) = (f32x4){val[i][0], val[i][1], val[i][2], val[i][3]};
          } else {
            u32x2 hp, lp;
            pz_split4(val[i][0], val[i][1], val[i][2], val[i][3], hp, lp);
            *reinterpret_cast<u32x2*>(row + (q >> 1) * kPLS + ps * 16 + (q & 1) * 8) = hp;
            *reinterpret_cast<u32x2*>(row + (G + (q >> 1)) * kPLS + ps * 16 + (q & 1) * 8) = lp;
          }
        }
      }
      if (htid < kPSL && y >= 0 && y < p.H) reinterpret_cast<float*>(smem + Z::DEPTH)[(y & (kDR - 1)) * kPSL + htid] = dval;
    };
#pragma unroll
    for (int j = 0; j < kPF; ++j) {
      dvalr[j] = 0.f;
      issue(j, valr[j], dvalr[j]);                               // rows 0 .. kPF - 1 (rows >= H load nothing and give zeros)
    }
    const int steps = p.H + 4 * kLag;
    for (int t0 = 0; t0 < steps; t0 += kPF) {
#pragma unroll
      for (int j = 0; j < kPF; ++j) {
        const int t = t0 + j;
        if (t < steps) {                                         // (wave-uniform; every wave of the workgroup runs `steps` barriers)
          if (t <= p.H) commit(t, valr[j], dvalr[j]);            // row H = the zero row below the image
          if (t + kPF <= p.H) issue(t + kPF, valr[j], dvalr[j]);
          __syncthreads();
        }
      }
    }
  }
}

template <int G, bool F32>
__global__ __launch_bounds__(kThreads, 2) void propz_kernel(PropzParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_items = p.B * p.nstrip;
  const int item0 = (int)blockIdx.x, item_step = (int)gridDim.x;
  // Waves are dealt to the four SIMDs round robin.  The image-guided net (G = 1: layer 1 has 54 matrix instructions per row) keeps
  // layer 1 on wave 0 and four helper waves; the feature-guided nets (layer 1: 216 at G = 5) split it over wave 0 and wave 7 (SIMD 3,
  // beside the layer-4 wave, the lightest) -- measured per net, 64 views: G = 1 0.96 ms either way round (1.04 with the split),
  // G = 5 0.41 -> 0.345 ms (128 x 160) and 0.207 -> 0.163 ms (64 x 80) with it.
  if constexpr (G == 1) {
    if (wave == 0) pz_matrix_role<G, 1, F32, 0, 2>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 1) pz_matrix_role<G, 2, F32, 0, 2>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 2) pz_matrix_role<G, 3, F32, 0, 2>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 3) pz_matrix_role<G, 4, F32, 0, 1>(p, smem, lane, item0, item_step, n_items);
    else pz_helper_role<G, F32, 256>(p, smem, tid - 256, item0, item_step, n_items);
  } else {
    if (wave == 0) pz_matrix_role<G, 1, F32, 0, 1>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 1) pz_matrix_role<G, 2, F32, 0, 2>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 2) pz_matrix_role<G, 3, F32, 0, 2>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 3) pz_matrix_role<G, 4, F32, 0, 1>(p, smem, lane, item0, item_step, n_items);
    else if (wave == 7) pz_matrix_role<G, 1, F32, 1, 1>(p, smem, lane, item0, item_step, n_items);
    else pz_helper_role<G, F32, 192>(p, smem, tid - 256, item0, item_step, n_items);
  }
}

template <int G, bool F32>
int launch_g(const PropzParams& p, hipStream_t s) {
  static bool attr_set[64] = {false};      // per device: the dynamic-LDS opt-in is a per-device function attribute
  int dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "device ordinal %d", dev);
  if (!attr_set[dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)propz_kernel<G, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PZ<G>::LDS));
    attr_set[dev] = true;
  }
  const long long items = (long long)p.B * p.nstrip;
  const unsigned grid = v3d::persistent_grid(items, 1);
  {
    v3d::TimedScope ts(F32 ? "propagation_fused_f32" : "propagation_fused", s);
    propz_kernel<G, F32><<<grid, kThreads, PZ<G>::LDS, s>>>(p);
  }
  V3D_CHECK_LAUNCH("propz_kernel");
  return V3D_OK;
}

}  // namespace

// Host side of the fragment images (v3d_propagation_pack, costreg.hip): words of layer l in the order the kernel reads them.
//   K step ks, 16-row block cb, part (hi, lo), lane (kq = lane >> 4, m = lane & 15): row = output channel cb * 16 + m, the lane's
//   8 k values e = 0 .. 7:  layers 2..4: tap = ks, input channel = 8 kq + e;  layer 1: k8 = ks * 4 + kq, tap = k8 / G (zero
//   weights for tap >= 9), input channel = (k8 % G) * 8 + e.
size_t v3d::propz_image_words(int layer, int cinp) {
  const int G = cinp / 8, ks = layer == 0 ? (9 * G + 3) / 4 : 9, ncb = layer == 3 ? 1 : 2;
  return (size_t)ks * ncb * 2 * 64 * 4;
}

void v3d::propz_pack_layer(int layer, int cinp, int cin, int cout, const float* w_folded, unsigned* out, bool f32) {
  const int G = cinp / 8, KS = layer == 0 ? (9 * G + 3) / 4 : 9, ncb = layer == 3 ? 1 : 2;
  auto rne = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
  auto up = [](unsigned hb) { unsigned u = hb << 16; float f; memcpy(&f, &u, 4); return f; };
  for (int ks = 0; ks < KS; ++ks)
    for (int cb = 0; cb < ncb; ++cb)
      for (int lane = 0; lane < 64; ++lane) {
        const int kq = lane >> 4, co = cb * 16 + (lane & 15);
        unsigned hi[8], lo[8];
        for (int e = 0; e < 8; ++e) {
          int tap, ci;
          if (layer == 0) { const int k8 = ks * 4 + kq; tap = k8 / G; ci = (k8 % G) * 8 + e; }
          else { tap = ks; ci = 8 * kq + e; }
          float v = 0.f;
          if (tap < 9 && co < cout && ci < cin) v = w_folded[((size_t)co * cin + ci) * 9 + tap];      // [cout][cin][ky][kx]
          hi[e] = rne(v);
          lo[e] = rne(v - up(hi[e]));
          if (f32) memcpy(&hi[e], &v, 4);      // exact-fp32 image: part 0 = k slices 0..3, part 1 = 4..7 (slice e <-> channel 8 kq + e)
        }
        for (int part = 0; part < 2; ++part) {
          const unsigned* src = part ? lo : hi;
          unsigned* dst = out + ((((size_t)ks * ncb + cb) * 2 + part) * 64 + lane) * 4;
          for (int q = 0; q < 4; ++q) dst[q] = f32 ? hi[4 * part + q] : (src[2 * q] | (src[2 * q + 1] << 16));
        }
      }
}

int v3d::launch_propz(int cinp, bool f32, const float* feat, const float* depth, const int* iy, const int* ix, float* out,
                      const float* const w[4], const float* const bias[4], int B, int Cf, int H, int W, int h0, int w0, hipStream_t s) {
  PropzParams p;
  p.feat = feat; p.depth = depth; p.iy = iy; p.ix = ix; p.out = out;
  for (int l = 0; l < 4; ++l) { p.w[l] = reinterpret_cast<const u32x4*>(w[l]); p.bias[l] = bias[l]; }
  p.B = B; p.Cf = Cf; p.H = H; p.W = W; p.h0 = h0; p.w0 = w0;
  p.nstrip = (W + kTWO - 1) / kTWO;
  V3D_REQUIRE((long long)B * p.nstrip < (1ll << 31), V3D_ERR_BAD_SHAPE, "propagation: too many strips");
  if (cinp == 8) return f32 ? launch_g<1, true>(p, s) : launch_g<1, false>(p, s);
  if (cinp == 24) return f32 ? launch_g<3, true>(p, s) : launch_g<3, false>(p, s);
  if (cinp == 40) return f32 ? launch_g<5, true>(p, s) : launch_g<5, false>(p, s);
  return v3d::fail(V3D_ERR_UNSUPPORTED, "propagation: %d padded input channels (8, 24, 40)", cinp);
}
