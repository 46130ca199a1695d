// conv1 + conv2 of CostRegNet (mvsnet.py:137-138: ConvBnRelu3d(8 -> 16, stride 2), ConvBnRelu3d(16 -> 16)) as ONE depth march
// on split-bf16 matrix cores (round 4; V3D_C12_MARCH=0 selects the two tile kernels instead).  Only conv0, conv2 and conv4 are skip connections:
// conv1's output is consumed by conv2 alone, so it never has to exist in HBM.  Arithmetic and weight images are those of
// convg_bf16x2_kernel (costreg.hip: rows = 16 output channels, K = 32 = 4 x taps x 8 input channels with a zero fourth tap,
// hi*hi + hi*lo + lo*hi per product, fp32 accumulation); the two 8-channel chunks of conv2 are summed chunk 0 + chunk 1 (the
// per-layer kernel runs them through one accumulator: same products, another order).
//
//   * a workgroup owns a 4 x 28 (y, x) tile of the half-resolution grid and walks z.  Per step (one half-resolution plane):
//     two conv0 planes (13 x 61 slots with halo; even columns first, then odd ones, so that conv1's stride-2 taps read
//     consecutive slots) arrive through a ring of four LDS-DMA buffers; waves 0, 1 (conv1, six of the twelve 16-column blocks
//     of the 6 x 30 conv1 tile each) run them through the matrix pipe -- the odd plane 2p - 1 closes conv1 plane p - 1 (kz = 2)
//     and opens plane p (kz = 0), the even plane 2p adds kz = 1 -- and hand the finished plane's accumulators to waves 4..7,
//     which park it (bias, ReLU, zero outside the volume, split) in one of two LDS buffers during the next step, besides the
//     DMA and conv2's reduction / bias / ReLU / split / store; waves 2, 3 (conv2, one input-channel chunk each, all 7 blocks)
//     consume the plane parked one step earlier exactly as conv0z.hip consumes an input plane (three accumulator sets in
//     registers, 21 items of 2 ds_read_b128 -> 9 MFMAs).
//   * the (y, x) halo is the only recompute: conv1 on 6 x 30 for 4 x 28 (1.6x of a layer that is a third of the pair's MFMAs).
//
// Measured (cfg2, 64 views): 0.27 ms against 0.176 + 0.146 ms for the two tile kernels (cfg5: +1.8 % on the whole step).  Cycle
// counters (scripts/phase_conv0z.py --kernel conv12z): a step is ~4.4 k cycles in which the conv1 waves issue MFMAs for ~2.7 k, the
// conv2 waves for ~2.9 k, the helpers work ~2.9 k -- and every role waits ~1.2 k at the two barriers: three roles of varying length
// in lockstep pay the maximum at every barrier.  What did not move it: the park inside the conv1 waves (as a phase 0.270,
// scheduled between the even plane's MFMAs 0.264 with spills), the park as a helper phase between B' and B (0.275), two conv1 waves
// per SIMD with two helper waves (0.350: the helpers' 26 DMA issues per step become the longest role), de-interleaved tile
// columns (the stride-2 reads were not the limiter), prefetch distance 1 / 3, wave priorities, more z segments, an LDS counter
// in place of the second barrier (V3D_C12_FLAGS: 0.293).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "v3d_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct C12 {
  static constexpr int TH = 4, TW = 28;                        // conv2 output tile (half-resolution y, x)
  static constexpr int NB2 = TH * TW / 16;                     // 7 column blocks
  static constexpr int H1 = TH + 2, W1 = TW + 2;               // conv1 tile with conv2's halo: 6 x 30
  static constexpr int NPX1 = H1 * W1, NB1 = (NPX1 + 15) / 16; // 180 positions, 12 blocks (6 per conv1 wave)
  static constexpr int H0 = 2 * H1 + 1, W0 = 2 * W1 + 1;       // conv0 tile: 13 x 61 slots
  static constexpr int NSLOT0 = H0 * W0;                       // 793
  static constexpr int NPIECE = 13;                            // DMA instructions per (plane, hi | lo): 832 slots (the pad stays zero)
  static constexpr int HL0 = NPIECE * 1024, PLANE0 = 2 * HL0;  // 28 KB per plane
  static constexpr int R0 = 4;                                 // ring: the two planes of a step + the two of the next
  static constexpr int P1 = 32;                                // slot pitch of a conv1 row
  static constexpr int HL1 = H1 * P1 * 16;                     // one (chunk, hi | lo) array of a conv1 plane
  static constexpr int BUF1 = 4 * HL1;                         // [chunk 2][hi, lo]
  static constexpr int RED = 2 * NB2 * 1024;                   // [chunk][block][lane] f32x4
  static constexpr int RAW = NB1 * 1024;                       // a finished conv1 plane's accumulators on their way to the park
  static constexpr int LDS_BYTES = R0 * PLANE0 + 2 * BUF1 + RED + RAW + 16;     // + the "taken" counter
  static_assert(TH * TW % 16 == 0 && NB1 == 12 && NSLOT0 <= NPIECE * 64 && 2 * NPIECE == 26 && LDS_BYTES <= 160 * 1024, "geometry");
};

struct C12Params {
  const void* c0;      // conv0 output, split layout [n][hi, lo][D][H][W] 16-byte slots of 8 channels
  const void* w1;      // conv1 fragments [9 (kz, ky)][hi, lo][64 lanes][4 words]
  const void* w2;      // conv2 fragments [2 chunks][9][hi, lo][64 lanes][4 words]
  const float* b1;     // [16] folded BN biases
  const float* b2;     // [16]
  void* out;           // conv2 output, split layout [n][2 groups][hi, lo][D2][H2][W2]
  int n, D, H, W, D2, H2, W2, nty, ntx, nseg, seg_len, n_tasks;
};

__device__ __forceinline__ unsigned c12_pack(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
__device__ __forceinline__ void c12_split4(const float (&v)[4], u32x2& hi, u32x2& lo) {
  const unsigned h01 = c12_pack(v[0], v[1]), h23 = c12_pack(v[2], v[3]);
  hi = (u32x2){h01, h23};
  lo = (u32x2){c12_pack(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u)),
               c12_pack(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u))};
}

// f(integral_constant<int, I>) for I = B .. E - 1, fully unrolled with compile-time indices
template <int B, int E, class F>
__device__ __forceinline__ void c12_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    c12_static_for<B + 1, E>(f);
  }
}
#ifdef V3D_PHASE_TIMING
// developer build only: wave 0 (conv1: marks 0-2), wave 2 (conv2: 3-5) and wave 4 (helper: 6, 7) of a workgroup write their cycle counts
__device__ unsigned long long g_c12_phase[8 * 1024];
#define C12_PHASE_DECL long long ph_t = __builtin_readcyclecounter(); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define C12_PHASE_MARK(i) do { const long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#define C12_PHASE_FLUSH(lo, hi) do { if (lane == 0 && blockIdx.x < 1024) for (int i_ = lo; i_ <= hi; ++i_) g_c12_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_]; } while (0)
#else
#define C12_PHASE_DECL
#define C12_PHASE_MARK(i)
#define C12_PHASE_FLUSH(lo, hi)
#endif
#ifndef V3D_C12_ABLATE
#define V3D_C12_ABLATE 0     // developer ablations: 1 no conv2 MFMAs, 2 no conv1 MFMAs, 3 no DMA, 4 no conv2 epilogue, 5 no conv1 park
#endif
#ifndef V3D_C12_FLAGS
#define V3D_C12_FLAGS 0      // developer A/B, 1: the second barrier of a step (B': "the helpers have read `raw` and `red`") as an LDS
#endif                       // counter the writers poll instead of a workgroup barrier -- correct, and SLOWER: 0.293 against 0.270 ms
// The counter counts helper waves that have finished the reads of their round (4 per round, monotonic over the kernel).  A poll
// that does not see its target within 2^16 tries gives up (wrong results, which the tests catch, instead of a hung GPU).
__device__ __forceinline__ void c12_wait_taken(const volatile unsigned* cnt, unsigned target) {
  for (int i = 0; i < (1 << 16); ++i) {
    if (*cnt >= target) return;
    __builtin_amdgcn_s_sleep(1);
  }
}
#ifndef V3D_C12_PRE
#define V3D_C12_PRE 2        // B fragments this many items ahead of their MFMAs (as conv0z.hip)
#endif

__global__ __launch_bounds__(512, 2) void conv12z_kernel(C12Params p, const float* __restrict__ b1_r, const float* __restrict__ b2_r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, jn = lane & 15;
  unsigned char* const ring0 = smem;                                    // [R0][hi, lo][NPIECE KB]
  unsigned char* const buf1 = smem + C12::R0 * C12::PLANE0;             // [2][chunk][hi, lo][H1][P1] slots
  f32x4* const red = reinterpret_cast<f32x4*>(buf1 + 2 * C12::BUF1);    // [chunk][block][lane]
  f32x4* const raw = red + C12::RED / 16;                               // [block][lane]
  unsigned* const taken = reinterpret_cast<unsigned*>(raw + C12::RAW / 16);
  if (tid == 0) *taken = 0u;
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  // the pad columns of the conv1 buffers (x = 30, 31: read by the idle fourth x tap) are never written: finite from the start
  for (int i = tid; i < 2 * C12::BUF1 / 16; i += 512) reinterpret_cast<u32x4*>(buf1)[i] = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  const size_t HW = (size_t)p.H * p.W, DHW = (size_t)p.D * HW;
  const size_t HW2 = (size_t)p.H2 * p.W2, DHW2 = (size_t)p.D2 * HW2;
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n_tasks);
  C12_PHASE_DECL;
  struct Task { int n, oy0, ox0, z0, z1, nsteps; };
  auto decode = [&](int t) __attribute__((always_inline)) {
    Task q;
    const int tx = t % p.ntx; t /= p.ntx;
    const int ty = t % p.nty; t /= p.nty;
    const int seg = t % p.nseg;
    q.n = t / p.nseg;
    q.oy0 = ty * C12::TH; q.ox0 = tx * C12::TW;
    q.z0 = seg * p.seg_len;
    q.z1 = min(q.z0 + p.seg_len, p.D2);
    q.nsteps = q.z1 - q.z0 + 5;
    return q;
  };
  // Step s of a task (segment [z0, z1) of conv2 planes), P = z0 - 1 + s:
  //   conv1 waves: odd conv0 plane 2P - 1 (kz = 2 -> conv1 plane P - 1, kz = 0 -> plane P), even conv0 plane 2P (kz = 1 -> plane
  //                P); behind B'(s) the accumulators of the finished plane P - 1 go to `raw` (planes z0 - 1 .. z1 in steps
  //                1 .. len + 2).
  //   helpers:     between B(s) and B'(s): stores of the conv2 plane taken last round, DMA of the planes of step s + 1 into the
  //                ring slots step s - 1 used, the park of the conv1 plane `raw` holds (finished in step s - 1: bias, ReLU, zeros
  //                outside the volume, split -> conv1 buffer s & 1), the partial sums of the conv2 plane step s - 1 completed.
  //   conv2 waves: the conv1 plane parked in step s - 1 (plane Q = P - 3, buffer (s - 1) & 1) -> conv2 planes Q + 1, Q, Q - 1;
  //                conv2 plane Q - 1 is complete: partial sums into `red` behind B'(s).
  // Barriers as in conv0z.hip: B(s) = __syncthreads at the top of a step (planes of step s landed; `raw`, the conv1 buffer and
  // `red` of step s - 1 complete); B'(s) = a bare s_barrier (the helpers have read `raw` and `red`).
  // Roles: waves 0, 1 conv1 (six of the twelve blocks each), waves 2, 3 conv2 (one input-channel chunk each), waves 4..7 DMA,
  // the park of the finished conv1 plane and conv2's epilogue -- VALU work that interleaves freely with the matrix waves'
  // MFMAs on the same SIMD, which a wave cannot do with its own MFMAs (it issues in order).
  if (wave8 < 2) {
    // ================================ conv1: blocks 6 w .. 6 w + 5 of the 6 x 30 tile ================================
    const int w = wave8;
    bf16x8 a_hi[9], a_lo[9];
    {
      const u32x4* wq = reinterpret_cast<const u32x4*>(p.w1) + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a_hi[k] = __builtin_bit_cast(bf16x8, wq[(k * 2) * 64]);
        a_lo[k] = __builtin_bit_cast(bf16x8, wq[(k * 2 + 1) * 64]);
      }
    }
    // B operand of block b, column jn (conv1 position q = (y1, x1)), x tap kq: conv0 slot (2 y1 + ky, 2 x1 + kq)
    unsigned boff[6];      // B operand of block b, column jn (conv1 position q = (y1, x1)), x tap kq: conv0 slot (2 y1 + ky, 2 x1 + kq)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int q = min(16 * (6 * w + b) + jn, C12::NPX1 - 1);
      const int y1 = q / C12::W1, x1 = q % C12::W1;
      // (a tile row holds its even columns first, then its odd ones -- see the DMA lanes: column 2 x1 + kq is slot x1 + kq / 2 of
      // its parity's run, so the 16 column lanes of a tap read 16 consecutive slots instead of every other one)
      boff[b] = (unsigned)(((2 * y1) * C12::W0 + ((kq & 1) ? C12::W1 + 1 : 0) + x1 + (kq >> 1)) * 16);
    }
    f32x4 acc_prev[6], acc_cur[6];
    unsigned gbase = 0;                                                // helper rounds of the tasks before this one
#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
#pragma unroll
      for (int b = 0; b < 6; ++b) acc_prev[b] = acc_cur[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int s = 0; s < q.nsteps; ++s) {
        const int P = q.z0 - 1 + s;
        __syncthreads();                                               // B(s)
        C12_PHASE_MARK(0);
        const int zo = 2 * P - 1, ze = 2 * P;
        const bool odd_ok = zo >= 0 && zo < p.D, even_ok = ze >= 0 && ze < p.D;
        const bool close_prev = s >= 1 && s <= q.z1 - q.z0 + 2;        // conv1 plane P - 1 in [z0 - 1, z1] is parked this step
        const bool open_cur = s <= q.z1 - q.z0 + 1;                    // conv1 plane P in [z0 - 1, z1] is still to be built
        const unsigned char* const ro = ring0 + ((2 * s) & 3) * C12::PLANE0;
        const unsigned char* const re = ring0 + ((2 * s + 1) & 3) * C12::PLANE0;
        // ---- odd plane: kz = 2 closes conv1 plane P - 1, kz = 0 opens plane P
        constexpr int NI1 = 18, kPre = V3D_C12_PRE;
        if (odd_ok && V3D_C12_ABLATE != 2) {
          u32x4 bh_[NI1], bl_[NI1];
          auto load = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / 6, b = i % 6;
            const unsigned bo = boff[b] + ky * (C12::W0 * 16);
            bh_[i] = *reinterpret_cast<const u32x4*>(ro + bo);
            bl_[i] = *reinterpret_cast<const u32x4*>(ro + bo + C12::HL0);
          };
          auto item = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / 6, b = i % 6;
            if constexpr (i + kPre < NI1) load(std::integral_constant<int, i + kPre>{});
            const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bh_[i]), b_lo = __builtin_bit_cast(bf16x8, bl_[i]);
            acc_prev[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[6 + ky], b_hi, acc_prev[b], 0, 0, 0);
            acc_cur[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[0 + ky], b_hi, acc_cur[b], 0, 0, 0);
            acc_prev[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[6 + ky], b_lo, acc_prev[b], 0, 0, 0);
            acc_cur[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[0 + ky], b_lo, acc_cur[b], 0, 0, 0);
            acc_prev[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[6 + ky], b_hi, acc_prev[b], 0, 0, 0);
            acc_cur[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[0 + ky], b_hi, acc_cur[b], 0, 0, 0);
            if constexpr (i + kPre < NI1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
          };
          c12_static_for<0, kPre>(load);
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * kPre, 0);
          c12_static_for<0, NI1>(item);
        }
        C12_PHASE_MARK(1);
        // ---- even plane: kz = 1 of conv1 plane P
        if (even_ok && open_cur && V3D_C12_ABLATE != 2) {
          u32x4 bh_[NI1], bl_[NI1];
          auto load = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / 6, b = i % 6;
            const unsigned bo = boff[b] + ky * (C12::W0 * 16);
            bh_[i] = *reinterpret_cast<const u32x4*>(re + bo);
            bl_[i] = *reinterpret_cast<const u32x4*>(re + bo + C12::HL0);
          };
          auto item = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / 6, b = i % 6;
            if constexpr (i + 2 * kPre < NI1) load(std::integral_constant<int, i + 2 * kPre>{});
            const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bh_[i]), b_lo = __builtin_bit_cast(bf16x8, bl_[i]);
            acc_cur[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[3 + ky], b_hi, acc_cur[b], 0, 0, 0);
            acc_cur[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[3 + ky], b_lo, acc_cur[b], 0, 0, 0);
            acc_cur[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[3 + ky], b_hi, acc_cur[b], 0, 0, 0);
            if constexpr (i + 2 * kPre < NI1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          };
          c12_static_for<0, 2 * kPre>(load);
          __builtin_amdgcn_sched_group_barrier(0x100, 4 * kPre, 0);
          c12_static_for<0, NI1>(item);
        }
        C12_PHASE_MARK(2);
        if (V3D_C12_FLAGS) c12_wait_taken(taken, 4u * (gbase + (unsigned)s + 1u));
        else asm volatile("s_barrier" ::: "memory");                   // B'(s): `raw` has been read
        C12_PHASE_MARK(3);
        // ---- conv1 plane P - 1 is complete since the odd plane: its accumulators go to the helper waves, which park it during
        // the next step (bias, ReLU, zero padding, split: ~30 VALU instructions per block -- in this wave, which issues in order,
        // they held up the even plane's MFMAs by ~800 cycles a step: 0.264 ms; as a helper phase between B' and B 0.275)
        if (close_prev && V3D_C12_ABLATE != 5) {
#pragma unroll
          for (int b = 0; b < 6; ++b) raw[(6 * w + b) * 64 + lane] = acc_prev[b];
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) { acc_prev[b] = acc_cur[b]; acc_cur[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
      __syncthreads();                                                 // B(nsteps)
      if (!V3D_C12_FLAGS) asm volatile("s_barrier" ::: "memory");      // B'(nsteps)
      gbase += (unsigned)q.nsteps + 1u;
    }
    if (wave8 == 0) C12_PHASE_FLUSH(0, 3);
  } else if (wave8 < 4) {
    // ================================ conv2: input-channel chunk c, all 7 blocks ================================
    const int c = wave8 - 2;
    bf16x8 a_hi[9], a_lo[9];
    {
      const u32x4* wq = reinterpret_cast<const u32x4*>(p.w2) + (size_t)c * (9 * 2 * 64) + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a_hi[k] = __builtin_bit_cast(bf16x8, wq[(k * 2) * 64]);
        a_lo[k] = __builtin_bit_cast(bf16x8, wq[(k * 2 + 1) * 64]);
      }
    }
    unsigned boff[C12::NB2];
#pragma unroll
    for (int b = 0; b < C12::NB2; ++b) {
      const int q = 16 * b + jn, y2 = q / C12::TW, x2 = q % C12::TW;
      boff[b] = (unsigned)(c * 2 * C12::HL1 + (y2 * C12::P1 + x2 + kq) * 16);
    }
    f32x4 acc[3][C12::NB2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int b = 0; b < C12::NB2; ++b) acc[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned gbase = 0;
#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
      auto step = [&](int s, auto u_c) __attribute__((always_inline)) {
        constexpr int U = decltype(u_c)::value;                        // = s % 3: accumulator rotation
        constexpr int A0 = (U + 1) % 3, A1 = U, A2 = (U + 2) % 3;      // conv2 planes Q + 1, Q, Q - 1
        const int Q = q.z0 - 4 + s;                                    // the conv1 plane parked in step s - 1
        const bool valid = s >= 3 && Q >= 0 && Q < p.D2 && Q <= q.z1;
        __syncthreads();                                               // B(s)
        if (valid && V3D_C12_ABLATE != 1) {
          const unsigned char* const rb = buf1 + ((s - 1) & 1) * C12::BUF1;
          constexpr int NI = 3 * C12::NB2, kPre = V3D_C12_PRE;
          u32x4 bh_[NI], bl_[NI];
          auto load = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / C12::NB2, b = i % C12::NB2;
            const unsigned bo = boff[b] + ky * (C12::P1 * 16);
            bh_[i] = *reinterpret_cast<const u32x4*>(rb + bo);
            bl_[i] = *reinterpret_cast<const u32x4*>(rb + bo + C12::HL1);
          };
          auto item = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / C12::NB2, b = i % C12::NB2;
            if constexpr (i + kPre < NI) load(std::integral_constant<int, i + kPre>{});
            {
              const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bh_[i]), b_lo = __builtin_bit_cast(bf16x8, bl_[i]);
              const f32x4 c0 = ky == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[A0][b];
              acc[A0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[0 * 3 + ky], b_hi, c0, 0, 0, 0);
              acc[A1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[1 * 3 + ky], b_hi, acc[A1][b], 0, 0, 0);
              acc[A2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[2 * 3 + ky], b_hi, acc[A2][b], 0, 0, 0);
              acc[A0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[0 * 3 + ky], b_lo, acc[A0][b], 0, 0, 0);
              acc[A1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[1 * 3 + ky], b_lo, acc[A1][b], 0, 0, 0);
              acc[A2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[2 * 3 + ky], b_lo, acc[A2][b], 0, 0, 0);
              acc[A0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[0 * 3 + ky], b_hi, acc[A0][b], 0, 0, 0);
              acc[A1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[1 * 3 + ky], b_hi, acc[A1][b], 0, 0, 0);
              acc[A2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[2 * 3 + ky], b_hi, acc[A2][b], 0, 0, 0);
            }
            if constexpr (i + kPre < NI) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);                                   // the item's MFMAs
          };
          c12_static_for<0, kPre>(load);
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * kPre, 0);
          c12_static_for<0, NI>(item);
        } else {
#pragma unroll
          for (int b = 0; b < C12::NB2; ++b) acc[A0][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        C12_PHASE_MARK(4);
        if (V3D_C12_FLAGS) c12_wait_taken(taken, 4u * (gbase + (unsigned)s + 1u));
        else asm volatile("s_barrier" ::: "memory");                   // B'(s): the helpers have read `red`
        C12_PHASE_MARK(5);
        // conv2 plane Q - 1 has its three taps (from conv1 planes Q - 2, Q - 1, Q; the absent ones are zero planes)
#pragma unroll
        for (int b = 0; b < C12::NB2; ++b) red[(c * C12::NB2 + b) * 64 + lane] = acc[A2][b];
      };
#pragma unroll 1
      for (int s = 0; s < q.nsteps; s += 3) {
        step(s, std::integral_constant<int, 0>{});
        if (s + 1 < q.nsteps) step(s + 1, std::integral_constant<int, 1>{});
        if (s + 2 < q.nsteps) step(s + 2, std::integral_constant<int, 2>{});
      }
      __syncthreads();                                                 // B(nsteps)
      if (!V3D_C12_FLAGS) asm volatile("s_barrier" ::: "memory");      // B'(nsteps)
      gbase += (unsigned)q.nsteps + 1u;
      // (the accumulator rotation restarts at U = 0 with the next task: every slot is re-initialised by its first use --
      // A0 starts from zero at ky == 0 or is cleared; A1, A2 of the first two steps only ever hold cleared values)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int b = 0; b < C12::NB2; ++b) acc[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (c == 0) C12_PHASE_FLUSH(4, 5);
  } else {
    // ================================ helpers: DMA quarter h, conv1 park blocks 3 h .. 3 h + 2, conv2 epilogue blocks h, h + 4 ================================
    const int h = wave8 - 4;
    float bias1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float c0 = b1_r[r], c1 = b1_r[4 + r], c2 = b1_r[8 + r], c3 = b1_r[12 + r];
      bias1[r] = kq == 0 ? c0 : kq == 1 ? c1 : kq == 2 ? c2 : c3;
    }
    // conv1 park: blocks 3 h .. 3 h + 2; lane (kq, jn) holds channels 4 kq .. 4 kq + 3 of conv1 position q = (y1, x1): one 8-byte half
    // of its hi slot and of its lo slot (the positions past the 180th go to column 31 of the last row, which nobody reads)
    unsigned woff[3];
    int y1v[3], x1v[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int q0 = 16 * (3 * h + b) + jn;
      const bool dead = q0 >= C12::NPX1;
      const int qq = min(q0, C12::NPX1 - 1), y1 = qq / C12::W1, x1 = qq % C12::W1;
      y1v[b] = y1; x1v[b] = x1;
      woff[b] = (unsigned)((kq >> 1) * 2 * C12::HL1 + ((dead ? C12::H1 - 1 : y1) * C12::P1 + (dead ? C12::P1 - 1 : x1)) * 16 + (kq & 1) * 8);
    }
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float c0 = b2_r[r], c1 = b2_r[4 + r], c2 = b2_r[8 + r], c3 = b2_r[12 + r];
      bias[r] = kq == 0 ? c0 : kq == 1 ? c1 : kq == 2 ? c2 : c3;
    }
#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
      // DMA: this wave moves pieces h, h + 4, ... (6 or 7 of the 26 1-KB pieces of a plane: hi pieces 0..12, lo pieces 13..25).
      // Lane l of piece i holds slot 64 (i % 13) + l of the 13 x 61 tile; slots outside the tile / the volume stay as zeroed
      // which of this lane's conv1 positions lie inside the half-resolution volume (conv2's zero padding elsewhere)
      unsigned inside = 0;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int gy = q.oy0 - 1 + y1v[b], gx = q.ox0 - 1 + x1v[b];
        inside |= (gy >= 0 && gy < p.H2 && gx >= 0 && gx < p.W2 ? 1u : 0u) << b;
      }
      unsigned voff[7];
      unsigned long long vmask[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int piece = min(h + 4 * i, 2 * C12::NPIECE - 1), slot = 64 * (piece % C12::NPIECE) + lane;
        // slot -> (row, column): the row's 31 even columns, then its 30 odd ones (the conv1 waves read with a column stride of 2)
        const int row = slot / C12::W0, rs_ = slot % C12::W0;
        const int col = rs_ <= C12::W1 ? 2 * rs_ : 2 * (rs_ - C12::W1 - 1) + 1;
        const int gy = 2 * q.oy0 - 3 + row, gx = 2 * q.ox0 - 3 + col;
        const bool ok = slot < C12::NSLOT0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        voff[i] = ok ? (unsigned)((gy * p.W + gx) * 16) : 0u;
        vmask[i] = h + 4 * i < 2 * C12::NPIECE ? __ballot(ok) : 0ull;  // 0: no such piece / outside the volume -- not issued
      }
      const char* const in_c = reinterpret_cast<const char*>(p.c0) + ((size_t)q.n * 2) * DHW * 16;
      auto issue = [&](int z, int rs) __attribute__((always_inline)) {
        // pieces h + 4 i: hi half for piece < 13, lo half otherwise; LDS destination = plane base + piece KB
        const unsigned dst = smem_lds + (unsigned)rs * C12::PLANE0;
        const char* const bh = in_c + (size_t)z * HW * 16;
        const char* const bl = bh + DHW * 16;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int piece = h + 4 * i;                                 // wave-uniform
          const char* const base = piece < C12::NPIECE ? bh : bl;
          const unsigned d = dst + (unsigned)piece * 1024u;
          if (vmask[i] == 0 || V3D_C12_ABLATE == 3) continue;          // wave-uniform (the waits below are vmcnt(0): no counting)
          unsigned long long sv;
          unsigned m0v;
          asm volatile(
              "s_mov_b64 %[sv], exec\n\t"
              "s_mov_b32 %[m0v], m0\n\t"
              "s_mov_b32 m0, %[d]\n\t"
              "s_mov_b64 exec, %[k]\n\t"
              "global_load_lds_dwordx4 %[v], %[b]\n\t"
              "s_mov_b64 exec, %[sv]\n\t"
              "s_mov_b32 m0, %[m0v]"
              : [sv] "=&s"(sv), [m0v] "=&s"(m0v)
              : [d] "s"(d), [b] "s"(base), [v] "v"(voff[i]), [k] "s"(vmask[i])
              : "memory", "scc");
        }
      };
      auto plane_in = [&](int z) { return z >= 0 && z < p.D; };
      // planes of step s: odd 2 P - 1 -> ring slot (2 s) & 3, even 2 P -> slot (2 s + 1) & 3; a plane outside the volume is not
      // loaded (the conv1 waves skip it)
      auto issue_step = [&](int s) __attribute__((always_inline)) {
        if (s < q.nsteps) {
          const int P = q.z0 - 1 + s;
          if (plane_in(2 * P - 1)) issue(2 * P - 1, (2 * s) & 3);
          if (plane_in(2 * P)) issue(2 * P, (2 * s + 1) & 3);
        }
      };
      // epilogue lanes: blocks h and h + 4; lane (kq, jn) holds channels 4 kq .. 4 kq + 3 of position (y2, x2)
      int fsp[2];
      bool fok[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int b = h + 4 * k;
        const int qq = 16 * b + jn, y2 = qq / C12::TW, x2 = qq % C12::TW;
        const int gy = q.oy0 + y2, gx = q.ox0 + x2;
        fok[k] = b < C12::NB2 && gy < p.H2 && gx < p.W2;
        fsp[k] = gy * p.W2 + gx;
      }
      u32x2* const outs = reinterpret_cast<u32x2*>(p.out) + (((size_t)q.n * 2 + (kq >> 1)) * 2 * DHW2) * 2 + (kq & 1);
      f32x4 part[2][2];
      auto take = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b = h + 4 * k;
          if (b >= C12::NB2) continue;
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) part[k][cc] = red[(cc * C12::NB2 + b) * 64 + lane];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      auto finish = [&](int zo) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b = h + 4 * k;
          if (b >= C12::NB2) continue;
          const f32x4 v = part[k][0] + part[k][1];
          float val[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) val[r] = fmaxf(v[r] + bias[r], 0.f);
          u32x2 hi, lo;
          c12_split4(val, hi, lo);
          if (fok[k]) {
            const size_t sp = (size_t)zo * HW2 + fsp[k];
            outs[sp * 2] = hi;
            outs[(DHW2 + sp) * 2] = lo;
          }
        }
      };
      // prologue: the slots of the tile that lie outside the volume are never written by the DMA -- zeros from here on.  Every
      // helper wave zeroes exactly the ring pieces its own DMA writes (nobody reads the ring: all waves are past B'(nsteps) of
      // the last task), and waits for those stores before its first copy.
      {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rs = 0; rs < C12::R0; ++rs)
#pragma unroll
          for (int i = 0; i < 7; ++i)
            if (h + 4 * i < 2 * C12::NPIECE)
              *reinterpret_cast<u32x4*>(ring0 + rs * C12::PLANE0 + (h + 4 * i) * 1024 + lane * 16) = (u32x4){0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      issue_step(0);
      bool fin_prev = false;
      int zo_prev = 0;
#pragma unroll 1
      for (int s = 0; s <= q.nsteps; ++s) {
        // the planes of step s must have landed: everything this wave issued.  The stores of the previous round's epilogue are
        // OLDER than the copies (they are issued in front of them below), so they do not stretch this wait.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // B(s)
        C12_PHASE_MARK(7);
        if (fin_prev) finish(zo_prev);                                 // the sums taken in the last round: bias, ReLU, split, stores
        issue_step(s + 1);                                             // into the ring slots step s - 1 read
        // the conv1 plane that step s - 1 finished (z0 - 3 + s, for s - 1 in [1, len + 2]): `raw` -> conv1 buffer s & 1
        if (s >= 2 && s <= q.z1 - q.z0 + 3 && V3D_C12_ABLATE != 5) {
          const int P1 = q.z0 - 3 + s;
          const bool plane_in1 = P1 >= 0 && P1 < p.D2;
          unsigned char* const dst = buf1 + (s & 1) * C12::BUF1;
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const f32x4 a = raw[(3 * h + b) * 64 + lane];
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (plane_in1 && ((inside >> b) & 1u)) ? fmaxf(a[r] + bias1[r], 0.f) : 0.f;
            u32x2 hi, lo;
            c12_split4(v, hi, lo);
            *reinterpret_cast<u32x2*>(dst + woff[b]) = hi;
            *reinterpret_cast<u32x2*>(dst + woff[b] + C12::HL1) = lo;
          }
        }
        // conv2 plane completed by step s - 1: Q - 1 with Q = z0 - 4 + (s - 1)
        const int zo = q.z0 - 6 + s;
        const bool fin = s >= 1 && zo >= q.z0 && zo < q.z1 && V3D_C12_ABLATE != 4;
        if (fin) take();
        fin_prev = fin; zo_prev = zo;
        C12_PHASE_MARK(6);
        if (V3D_C12_FLAGS) {
          // the reads of `raw` and `red` have returned (park: values consumed above; take: explicit wait): one count per wave
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(taken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          asm volatile("s_barrier" ::: "memory");                      // B'(s)
        }
      }
      if (fin_prev) finish(zo_prev);
    }
    if (h == 0) C12_PHASE_FLUSH(6, 7);
  }
}

}  // namespace

int v3d::launch_conv12z(const void* c0, const float* w1, const float* w2, const float* b1, const float* b2, void* out, int n,
                        int D, int H, int W, hipStream_t s) {
  V3D_REQUIRE((long long)D * H * W * 16 < (1ll << 32), V3D_ERR_BAD_SHAPE, "conv1+conv2: volume too large for 32-bit plane offsets");
  C12Params p;
  p.c0 = c0; p.w1 = w1; p.w2 = w2; p.b1 = b1; p.b2 = b2; p.out = out;
  p.n = n; p.D = D; p.H = H; p.W = W;
  p.D2 = (D - 1) / 2 + 1; p.H2 = (H - 1) / 2 + 1; p.W2 = (W - 1) / 2 + 1;
  p.nty = (p.H2 + C12::TH - 1) / C12::TH; p.ntx = (p.W2 + C12::TW - 1) / C12::TW;
  int dev = 0, n_cu = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  const long long tiles = (long long)n * p.nty * p.ntx;
  long long best = -1;
  for (int nseg = 1; nseg <= p.D2; ++nseg) {
    const int len = (p.D2 + nseg - 1) / nseg;
    if ((long long)len * (nseg - 1) >= p.D2) continue;
    const long long rounds = (tiles * nseg + n_cu - 1) / n_cu;
    const long long cost = rounds * (len + 5) + 2 * rounds;
    if (best < 0 || cost < best) { best = cost; p.nseg = nseg; p.seg_len = len; }
  }
  if (const int nseg = v3d::option(v3d::kOptC12Nseg)) {        // developer A/B (v3d_set_option "c12_nseg")
    if (nseg >= 1 && nseg <= p.D2) { p.nseg = nseg; p.seg_len = (p.D2 + nseg - 1) / nseg; p.nseg = (p.D2 + p.seg_len - 1) / p.seg_len; }
  }
  const long long tasks = tiles * p.nseg;
  V3D_REQUIRE(tasks > 0 && tasks < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv1+conv2: bad grid");
  p.n_tasks = (int)tasks;
  static bool attr_set[64] = {false};
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "conv1+conv2: device ordinal %d", dev);
  if (!attr_set[dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv12z_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C12::LDS_BYTES));
    attr_set[dev] = true;
  }
  {
    v3d::TimedScope ts("costreg_conv12", s);
    conv12z_kernel<<<v3d::persistent_grid(tasks, 1), 512, C12::LDS_BYTES, s>>>(p, b1, b2);
  }
  V3D_CHECK_LAUNCH("conv12z_kernel");
  return V3D_OK;
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_conv12z_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  static unsigned long long h[8 * 1024];
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c12_phase), sizeof(h)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < 1024; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif
