// conv0 of CostRegNet (mvsnet.py:136: ConvBnRelu3d(32 -> 8), k3 p1) on split-bf16 matrix cores, written as a
// DEPTH MARCH (round 4).  Same arithmetic decomposition as conv0_bf16x2_kernel in costreg.hip -- every fp32 operand
// x = hi + lo (bf16 pairs), products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulation; MFMA rows =
// 2 x-shifts x 8 output channels, K = 4 x taps x 8 input channels, the host's weight image [chunk][kz*3+ky][hi, lo][lane][4]
// is the one that kernel reads -- but the work is laid out differently:
//
//   * A workgroup owns an 8 x 28 (y, x) tile of one view and WALKS z: every input plane is brought into LDS once and feeds
//     the three output planes it touches (kz = 0, 1, 2), whose accumulators stay in registers.  The 4 x 8 x 28 tiles of the
//     old kernel staged 6 x 10 x 30 voxels for 4 x 8 x 28 outputs (2.0x); here it is 10 x 30 for 8 x 28 per plane (1.34x).
//   * The four waves are specialised by INPUT CHANNEL CHUNK (8 channels each): a wave's 18 weight fragments (9 (kz, ky)
//     x hi / lo) live in its registers for the whole kernel -- no weight traffic at all in the loop -- and it streams its
//     own chunk of the input through its own ring of LDS-DMA buffers (global_load_lds_dwordx4, two tile rows per
//     instruction, three planes deep, waited for with vmcnt): no staging registers, no commit, no barrier on the input
//     side.  The four partial sums of an output plane meet in LDS once per plane (fixed order: deterministic).
//   * Columns of an MFMA are 16 FLATTENED (row, x-pair) positions of the tile (8 rows x 14 pairs = 7 full blocks), so all
//     16 columns are outputs (the row-per-block mapping of the old kernel used 14 of 16).
//
// Per plane and wave: 42 ds_read_b128 feed 189 MFMAs.  LDS: 4 x 3 x 10 KB rings + 28 KB reduction buffer = 148 KB, one
// workgroup per CU, grid = CUs, tasks (view, z segment, tile) in XCD-contiguous order so that the tiles of one view's
// z range -- which share halos -- run side by side on one XCD's L2.
#include <type_traits>
#include <utility>

#include "v3d_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct CZ {
  static constexpr int TH = 8, TW = 28, NP = TW / 2;        // output tile; x pairs per row
  static constexpr int NBLK = TH * NP / 16;                 // 7 column blocks of 16 flattened (row, pair) positions
  static constexpr int IH = TH + 2, IWS = 32;               // input rows; 16-byte slots per LDS row (30 used)
  static constexpr int NPIECE = IH / 2;                     // LDS-DMA instructions per (plane, hi | lo): two rows each
  static constexpr int HL_BYTES = NPIECE * 1024;
  static constexpr int PLANE_BYTES = 2 * HL_BYTES;          // hi rows, then lo rows
  static constexpr int NISSUE = 2 * NPIECE;                 // DMA instructions per plane and wave
  static constexpr int R = 3;                               // ring depth (planes)
  static constexpr int RING_BYTES = R * PLANE_BYTES;        // per wave
  static constexpr int RED_BYTES = 4 * NBLK * 1024;         // [chunk][block][lane] f32x4
  static constexpr int LDS_BYTES = 4 * RING_BYTES + RED_BYTES;
  static_assert(TH * NP % 16 == 0 && IH % 2 == 0 && TW + 2 <= IWS - 2, "geometry (two pad slots per row carry the keep-alive lanes)");
  static_assert(NISSUE == 10, "the vmcnt immediates below");
};

struct CZParams {
  const void* in;      // split volume [n][4 chunks][hi, lo][D][H][W] 16-byte slots
  const void* wp;      // [4 chunks][9 (kz, ky)][hi, lo][64 lanes][4 words]
  const float* bias;   // [8]
  void* out;           // split activation [n][hi, lo][D][H][W] 16-byte slots (8 channels)
  int n, D, H, W, nty, ntx, nseg, seg_len, n_tasks;
};

__device__ __forceinline__ unsigned cz_bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned cz_pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
__device__ __forceinline__ float cz_lane_select(int cond, float a, float b) {
  const unsigned m = 0u - (unsigned)(cond != 0);
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

// f(integral_constant<int, I>) for I = B .. E - 1, fully unrolled with compile-time indices
template <int B, int E, class F>
__device__ __forceinline__ void cz_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    cz_static_for<B + 1, E>(f);
  }
}

#ifdef V3D_PHASE_TIMING
// developer build only: every wave accumulates the cycles between marks in registers; wave 0 (matrix role, marks 0-3) and
// wave 4 (helper role, marks 2, 4-7) of a workgroup write them out
__device__ unsigned long long g_cz_phase[8 * 1024];
#define CZ_PHASE_DECL long long ph_t = __builtin_readcyclecounter(); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CZ_PHASE_MARK(i) do { const long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#define CZ_PHASE_FLUSH do { if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x < 1024) for (int i_ = 0; i_ < 8; ++i_) if ((threadIdx.x == 0) == (i_ == 0 || i_ == 1 || i_ == 3)) g_cz_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_]; } while (0)
#else
#define CZ_PHASE_DECL
#define CZ_PHASE_MARK(i)
#define CZ_PHASE_FLUSH
#endif

#ifndef V3D_CZ_MPRIO
#define V3D_CZ_MPRIO 0       // s_setprio of the matrix waves / of the helper waves (developer A/B)
#endif
#ifndef V3D_CZ_HPRIO
#define V3D_CZ_HPRIO 0
#endif
#ifndef V3D_CZ_ROT
#define V3D_CZ_ROT 12        // tile row r is stored rotated by V3D_CZ_ROT * r slots (0 = off; 12: the column blocks that
#endif                       // straddle two rows then read 16 distinct bank groups); costs 14 address registers
#ifndef V3D_CZ_PRE
#define V3D_CZ_PRE 2         // B fragments this many items ahead of their MFMAs
#endif
#ifndef V3D_CZ_ABLATE
#define V3D_CZ_ABLATE 0      // developer ablations: 1 no MFMAs, 2 no DMA, 3 no reduction / finalize
#endif

// F32 (exact-fp32 operands, V3D_PRECISION_FP32): the input is the fp32 channel-last volume of v3d_psv_variance_cl8 -- the split
// layout's addressing, the "hi" rows holding channels 0..3 and the "lo" rows channels 4..7 of a group as floats -- the products
// run on v_mfma_f32_16x16x4_f32 (K = 4 x taps of ONE channel per instruction, 8 instructions per (kz, ky) where the bf16 path
// has 3), the output is the fp32 [n, 8, D, H, W] tensor the exact-fp32 per-layer kernels continue from.
template <bool F32>
__global__ __launch_bounds__(512, 2) void conv0z_kernel(CZParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = wave8 >= 4;                                      // waves 4..7: DMA + reduction / epilogue
  const int wave = wave8 & 3;                                          // = input channel chunk (both roles)
  const int kq = lane >> 4, jn = lane & 15;
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned ring_lds = smem_lds + (unsigned)wave * CZ::RING_BYTES;      // what M0 carries
  unsigned char* const ring = smem + wave * CZ::RING_BYTES;
  f32x4* const red = reinterpret_cast<f32x4*>(smem + 4 * CZ::RING_BYTES);

  const size_t HW = (size_t)p.H * p.W, DHW = (size_t)p.D * HW;
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n_tasks);
  CZ_PHASE_DECL;
  struct Task { int n, oy0, ox0, z0, nsteps; };
  auto decode = [&](int t) __attribute__((always_inline)) {
    Task q;                                                            // (view, z segment, y tile, x tile), x fastest
    const int tx = t % p.ntx; t /= p.ntx;
    const int ty = t % p.nty; t /= p.nty;
    const int seg = t % p.nseg;
    q.n = t / p.nseg;
    q.oy0 = ty * CZ::TH; q.ox0 = tx * CZ::TW;
    q.z0 = seg * p.seg_len;
    q.nsteps = min(q.z0 + p.seg_len, p.D) - q.z0 + 2;                  // input planes z0 - 1 .. z1
    return q;
  };
  // Barrier protocol of a task (both roles execute it; step s handles input plane zi = z0 - 1 + s):
  //   B(s):  the plane of step s has landed in ring slot s % 3 (the helper waited for it); the matrix waves are done with
  //          step s - 1: its ring slot is free, `red` holds the partial sums of the out plane that step completed.
  //   B'(s): (near the end of the matrix waves' step, at the end of the helpers' round) the helpers have read those partial
  //          sums: `red` is free for the ones of step s, which the matrix waves write behind it.
  //   after the last step: B(nsteps), B'(nsteps) hand over the last out plane.

  if (!helper) {
    if (V3D_CZ_MPRIO) __builtin_amdgcn_s_setprio(V3D_CZ_MPRIO);
    // ================= matrix role: chunk `wave` of every input plane -> partial sums of 3 out planes =================
    // this wave's weight fragments: resident for the whole kernel (72 registers in both arithmetic types)
    bf16x8 a_hi[F32 ? 1 : 9], a_lo[F32 ? 1 : 9];
    float a32[F32 ? 9 : 1][8];                                         // F32: [kz * 3 + ky][channel]: lane (kq, m) = W[row m][x tap kq]
    if constexpr (F32) {
      const float* wq = reinterpret_cast<const float*>(p.wp) + (size_t)wave * (9 * 8 * 64) + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) a32[k][e] = wq[(k * 8 + e) * 64];
    } else {
      const u32x4* wq = reinterpret_cast<const u32x4*>(p.wp) + (size_t)wave * (9 * 2 * 64) + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a_hi[k] = __builtin_bit_cast(bf16x8, wq[(k * 2) * 64]);
        a_lo[k] = __builtin_bit_cast(bf16x8, wq[(k * 2 + 1) * 64]);
      }
    }
    // B operand of block b, column jn, k group kq: slot (y, 2 xp + kq) of the tile row y + ky (ky: immediate offset)
    unsigned boff[V3D_CZ_ROT ? 3 : 1][CZ::NBLK];
#pragma unroll
    for (int b = 0; b < CZ::NBLK; ++b) {
      const int q = 16 * b + jn, y = q / CZ::NP, xp = q % CZ::NP;
      if (V3D_CZ_ROT) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          boff[V3D_CZ_ROT ? ky : 0][b] = (unsigned)(wave * CZ::RING_BYTES + ((y + ky) * CZ::IWS + ((2 * xp + kq + V3D_CZ_ROT * (y + ky)) & 31)) * 16);
      } else {
        boff[0][b] = (unsigned)(wave * CZ::RING_BYTES + (y * CZ::IWS + 2 * xp + kq) * 16);   // + the wave's ring: one VGPR per block,
      }                                                                                     // everything else is an immediate
    }
    f32x4 acc[3][CZ::NBLK];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int b = 0; b < CZ::NBLK; ++b) acc[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
      auto put_plain = [&](const f32x4& v, int b) __attribute__((always_inline)) {
        if (V3D_CZ_ABLATE != 3) red[(wave * CZ::NBLK + b) * 64 + lane] = v;
      };
      // U = s % 3 fixes the ring slot and the accumulator slots
      auto step = [&](int s, auto u_c) __attribute__((always_inline)) {
        constexpr int U = decltype(u_c)::value;
        constexpr int A0 = (U + 1) % 3, A1 = U, A2 = (U + 2) % 3;      // accumulators of out planes zi + 1, zi, zi - 1
        const int zi = q.z0 - 1 + s;
        const bool valid = zi >= 0 && zi < p.D;
        __syncthreads();                                               // B(s)
        CZ_PHASE_MARK(0);
        if (valid && V3D_CZ_ABLATE != 1) {
          const unsigned char* const rb = smem + U * CZ::PLANE_BYTES;
          // 21 (ky, block) items, each 2 ds_read_b128 -> 9 MFMAs; the B fragments run kPre items ahead of the MFMAs, the
          // scheduler is pinned to that order
          constexpr int NI = 3 * CZ::NBLK, kPre = V3D_CZ_PRE;
          u32x4 bh_[NI], bl_[NI];
          auto load = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / CZ::NBLK, b = i % CZ::NBLK;
            const unsigned bo = V3D_CZ_ROT ? boff[V3D_CZ_ROT ? ky : 0][b] : boff[0][b] + ky * (CZ::IWS * 16);
            bh_[i] = *reinterpret_cast<const u32x4*>(rb + bo);
            bl_[i] = *reinterpret_cast<const u32x4*>(rb + bo + CZ::HL_BYTES);
          };
          // Out plane zi - 1 (slot A2) is complete block by block during the ky = 2 items (block b after item 14 + b).  Its
          // partial sums leave for LDS behind the LAST B-fragment reads (issued with item NI - 1 - kPre): lgkmcnt counts in
          // order, so a store in front of a read would make the wait for that read a wait for the store.  Blocks 0 .. 4 go
          // in the shadow of the last two items' MFMAs, blocks 5 and 6 behind them.
          auto put = [&](int b) __attribute__((always_inline)) {
            if (V3D_CZ_ABLATE != 3) red[(wave * CZ::NBLK + b) * 64 + lane] = acc[A2][b];
          };
          auto item = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / CZ::NBLK, b = i % CZ::NBLK;
            if constexpr (i + kPre < NI) load(std::integral_constant<int, i + kPre>{});
            if constexpr (i == NI - 2) { put(0); put(1); put(2); }
            if constexpr (i == NI - 1) { put(3); put(4); }
            constexpr int NM = F32 ? 24 : 9;                           // MFMAs per item
            if constexpr (F32) {
              // one channel per instruction (k = the 4 x taps): channels 0..3 from the first half slot, 4..7 from the second;
              // kz = 0 is the first contribution to out plane zi + 1: its first product starts from zero
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float bv = __uint_as_float(e < 4 ? bh_[i][e & 3] : bl_[i][e & 3]);
                const f32x4 c0 = (ky == 0 && e == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[A0][b];
                acc[A0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32[0 * 3 + ky][e], bv, c0, 0, 0, 0);
                acc[A1][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32[1 * 3 + ky][e], bv, acc[A1][b], 0, 0, 0);
                acc[A2][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32[2 * 3 + ky][e], bv, acc[A2][b], 0, 0, 0);
              }
            } else {
              const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bh_[i]), b_lo = __builtin_bit_cast(bf16x8, bl_[i]);
              // kz = 0 is the first contribution to out plane zi + 1: its first product starts from zero
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
            if constexpr (i == NI - 2 && V3D_CZ_ABLATE != 3) {                                   // stores between the MFMAs
              __builtin_amdgcn_sched_group_barrier(0x008, NM / 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, NM / 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, NM / 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            } else if constexpr (i == NI - 1 && V3D_CZ_ABLATE != 3) {
              __builtin_amdgcn_sched_group_barrier(0x008, NM / 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, NM / 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, NM / 3, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);                                // the item's MFMAs
            }
          };
          cz_static_for<0, kPre>(load);
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * kPre, 0);
          cz_static_for<0, NI - 2>(item);
          // B'(s), in front of the first store into `red`: a bare s_barrier -- nothing of this wave's memory traffic has to be
          // complete here (the B fragments in flight stay in flight); it orders the stores below, which follow it in
          // program order, behind the helpers' reads of the previous plane's sums.  The helpers reach it after a whole
          // round of their own work (DMA issue + epilogue), normally long before this wave.
          asm volatile("s_barrier" ::: "memory");
          cz_static_for<NI - 2, NI>(item);
          put(5);
          put(6);
        } else {
          asm volatile("s_barrier" ::: "memory");                      // B'(s)
          if (!valid) {
            // (only the last step of the last segment, zi = D: out plane D - 1 is complete without it)
#pragma unroll
            for (int b = 0; b < CZ::NBLK; ++b) put_plain(acc[A2][b], b);
#pragma unroll
            for (int b = 0; b < CZ::NBLK; ++b) acc[A0][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
        CZ_PHASE_MARK(1);
      };
#pragma unroll 1
      for (int s = 0; s < q.nsteps; s += 3) {
        step(s, std::integral_constant<int, 0>{});
        if (s + 1 < q.nsteps) step(s + 1, std::integral_constant<int, 1>{});
        if (s + 2 < q.nsteps) step(s + 2, std::integral_constant<int, 2>{});
      }
      __syncthreads();                                                 // B(nsteps)
      asm volatile("s_barrier" ::: "memory");                          // B'(nsteps)
      CZ_PHASE_MARK(3);
    }
  } else {
    // ================= helper role: this chunk's LDS-DMA stream + a quarter of the reduction / epilogue =================
    if (V3D_CZ_HPRIO) __builtin_amdgcn_s_setprio(V3D_CZ_HPRIO);
    float bias[4];
    {
      float sbias[8];                                                  // wave-uniform addresses: scalar loads
#pragma unroll
      for (int r = 0; r < 8; ++r) sbias[r] = p.bias[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[r] = cz_lane_select(kq & 1, sbias[4 + r], sbias[r]);
    }
#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
      // DMA lanes: lane = (row of the piece, slot of the row); slots 30, 31 are never read and always load (the plane's first
      // slot), so every piece is issued whatever the tile and the vmcnt arithmetic is exact
      unsigned voff[CZ::NPIECE];
      unsigned long long vmask[CZ::NPIECE];
      {
        const int j = lane >> 5, col0 = lane & 31;
#pragma unroll
        for (int i = 0; i < CZ::NPIECE; ++i) {
          const int col = (col0 - V3D_CZ_ROT * (2 * i + j)) & 31;      // the tile column this lane's slot holds
          const int gx = q.ox0 - 1 + col;
          const int gy = q.oy0 - 1 + 2 * i + j;
          const bool ok = col < CZ::TW + 2 && gx >= 0 && gx < p.W && gy >= 0 && gy < p.H;
          voff[i] = ok ? (unsigned)((gy * p.W + gx) * 16) : 0u;
          vmask[i] = __ballot(ok || col >= CZ::TW + 2);
        }
      }
      // epilogue lanes: this wave finishes blocks w and w + 4; lane (kq, jn) holds channels 4 (kq & 1) .. + 3 of the voxel
      // (y, 2 xp + (kq >> 1)) -- one 8-byte half of its hi slot and of its lo slot
      int fsp[2];
      bool fok[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int b = wave + 4 * k;
        const int qq = 16 * b + jn, y = qq / CZ::NP, xp = qq % CZ::NP;
        const int gy = q.oy0 + y, gx = q.ox0 + 2 * xp + (kq >> 1);
        fok[k] = b < CZ::NBLK && gy < p.H && gx < p.W;
        fsp[k] = gy * p.W + gx;
      }
      const char* const in_c = reinterpret_cast<const char*>(p.in) + ((size_t)(q.n * 4 + wave) * 2) * DHW * 16;
      u32x2* const outs = reinterpret_cast<u32x2*>(p.out) + ((size_t)q.n * 2 * DHW) * 2 + (kq & 1);
      float* const out32 = reinterpret_cast<float*>(p.out) + (size_t)q.n * 8 * DHW;

      // one plane of this chunk -> ring slot `rs`: 5 pieces of hi rows, 5 of lo rows
      auto issue = [&](int z, int rs) __attribute__((always_inline)) {
        if (V3D_CZ_ABLATE == 2) return;
        const char* const bh = in_c + (size_t)z * HW * 16;
        const char* const bl = bh + DHW * 16;
        const unsigned dst = ring_lds + (unsigned)rs * CZ::PLANE_BYTES;
        unsigned long long sv;
        unsigned m0v;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_mov_b32 %[m0v], m0\n\t"
            "s_mov_b32 m0, %[dst]\n\t"
            "s_mov_b64 exec, %[k0]\n\t"
            "global_load_lds_dwordx4 %[v0], %[bh]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k1]\n\t"
            "global_load_lds_dwordx4 %[v1], %[bh]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k2]\n\t"
            "global_load_lds_dwordx4 %[v2], %[bh]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k3]\n\t"
            "global_load_lds_dwordx4 %[v3], %[bh]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k4]\n\t"
            "global_load_lds_dwordx4 %[v4], %[bh]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k0]\n\t"
            "global_load_lds_dwordx4 %[v0], %[bl]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k1]\n\t"
            "global_load_lds_dwordx4 %[v1], %[bl]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k2]\n\t"
            "global_load_lds_dwordx4 %[v2], %[bl]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k3]\n\t"
            "global_load_lds_dwordx4 %[v3], %[bl]\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "s_mov_b64 exec, %[k4]\n\t"
            "global_load_lds_dwordx4 %[v4], %[bl]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_mov_b32 m0, %[m0v]"
            : [sv] "=&s"(sv), [m0v] "=&s"(m0v)
            : [dst] "s"(dst), [bh] "s"(bh), [bl] "s"(bl), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]),
              [v3] "v"(voff[3]), [v4] "v"(voff[4]), [k0] "s"(vmask[0]), [k1] "s"(vmask[1]), [k2] "s"(vmask[2]),
              [k3] "s"(vmask[3]), [k4] "s"(vmask[4])
            : "memory", "scc");
      };
      const auto plane_ok = [&](int s) { const int z = q.z0 - 1 + s; return s < q.nsteps && z >= 0 && z < p.D; };

      // the partial sums of the out plane step `sp` completed (zo = z0 - 2 + sp): registers before B', epilogue after it
      f32x4 part[2][4];
      auto take = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b = wave + 4 * k;
          if (b >= CZ::NBLK) continue;                                 // wave-uniform
#pragma unroll
          for (int c = 0; c < 4; ++c) part[k][c] = red[(c * CZ::NBLK + b) * 64 + lane];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      auto finish = [&](int zo) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int b = wave + 4 * k;
          if (b >= CZ::NBLK) continue;
          const f32x4 v = ((part[k][0] + part[k][1]) + part[k][2]) + part[k][3];
          float val[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) val[r] = fmaxf(v[r] + bias[r], 0.f);
          // hi = RNE_bf16(x), lo = RNE_bf16(x - hi) on packed pairs (v_cvt_pk_bf16_f32: the same rounding as the shift
          // arithmetic of the other kernels for finite values)
          const unsigned h01 = cz_pack_bf16x2(val[0], val[1]), h23 = cz_pack_bf16x2(val[2], val[3]);
          const unsigned l01 = cz_pack_bf16x2(val[0] - __uint_as_float(h01 << 16), val[1] - __uint_as_float(h01 & 0xffff0000u));
          const unsigned l23 = cz_pack_bf16x2(val[2] - __uint_as_float(h23 << 16), val[3] - __uint_as_float(h23 & 0xffff0000u));
          if (fok[k]) {
            const size_t sp = (size_t)zo * HW + fsp[k];
            if constexpr (F32) {
              // fp32 [n, 8, D, H, W]: this lane's four channels of its voxel (the 16 lanes of a quarter cover every other x
              // of a row; the quarter with the other x parity fills the gaps of the same 128-byte lines)
#pragma unroll
              for (int r = 0; r < 4; ++r) out32[(size_t)(4 * (kq & 1) + r) * DHW + sp] = val[r];
            } else {
              outs[sp * 2] = (u32x2){h01, h23};
              outs[(DHW + sp) * 2] = (u32x2){l01, l23};
            }
          }
        }
      };

      // prologue: out-of-volume slots of the tile are never written by the DMA -- zeros from here on (the matrix waves are
      // past B'(nsteps) of the previous task: nobody reads the ring) -- then the planes of steps 0 and 1
      {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4* const rz = reinterpret_cast<u32x4*>(ring);
#pragma unroll
        for (int i = 0; i < CZ::RING_BYTES / 1024; ++i) rz[i * 64 + lane] = (u32x4){0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (plane_ok(0)) issue(q.z0 - 1, 0);
      if (plane_ok(1)) issue(q.z0, 1);
      CZ_PHASE_MARK(4);
#pragma unroll 1
      for (int s = 0; s <= q.nsteps; ++s) {
        // the plane of step s must have landed; at most the one of step s + 1 may be in flight (loads return in order, the
        // epilogue's stores only make the wait conservative)
        if (plane_ok(s + 1)) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CZ_PHASE_MARK(5);
        __syncthreads();                                               // B(s)
        CZ_PHASE_MARK(6);
        const bool fin = s >= 3 && V3D_CZ_ABLATE != 3;                 // step s - 1 >= 2 completed out plane z0 - 3 + s
        if (plane_ok(s + 2)) issue(q.z0 + 1 + s, (s + 2) % 3);         // the slot step s - 1 read
        if (fin) { take(); finish(q.z0 - 3 + s); }
        CZ_PHASE_MARK(7);
        __syncthreads();                                               // B'(s): `red` has been read
        CZ_PHASE_MARK(2);
      }
    }
  }
  CZ_PHASE_FLUSH;
}

}  // namespace

// conv0 + folded BN + ReLU of a batch of variance volumes: split-bf16 hand-off formats in and out (f32 = false), or the fp32
// channel-last volume in and the fp32 [n, 8, D, H, W] tensor out (f32 = true; `wimg` = the fp32 fragment image).
int v3d::launch_conv0z(bool f32, const void* in, const float* wimg, const float* bias, void* out, int n, int D, int H, int W,
                       hipStream_t s) {
  V3D_REQUIRE((long long)D * H * W * 16 < (1ll << 32), V3D_ERR_BAD_SHAPE, "conv0: volume too large for 32-bit plane offsets");
  CZParams p;
  p.in = in; p.wp = wimg; p.bias = bias; p.out = out;
  p.n = n; p.D = D; p.H = H; p.W = W;
  p.nty = (H + CZ::TH - 1) / CZ::TH; p.ntx = (W + CZ::TW - 1) / CZ::TW;
  int dev = 0, n_cu = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  // z segments: every task also walks the two halo planes of its segment, every CU runs ceil(tasks / CUs) tasks
  const long long tiles = (long long)n * p.nty * p.ntx;
  long long best = -1;
  for (int nseg = 1; nseg <= D; ++nseg) {
    const int len = (D + nseg - 1) / nseg;
    if ((long long)len * (nseg - 1) >= D) continue;           // an empty last segment
    const long long rounds = (tiles * nseg + n_cu - 1) / n_cu;
    const long long cost = rounds * (len + 2) + 2 * rounds;   // + task turnover
    if (best < 0 || cost < best) { best = cost; p.nseg = nseg; p.seg_len = len; }
  }
  const long long tasks = tiles * p.nseg;
  V3D_REQUIRE(tasks > 0 && tasks < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv0: bad grid");
  p.n_tasks = (int)tasks;
  static bool attr_set[64][2] = {{false}};
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "conv0: device ordinal %d", dev);
  if (!attr_set[dev][f32]) {
    V3D_CHECK_HIP(hipFuncSetAttribute(f32 ? (const void*)conv0z_kernel<true> : (const void*)conv0z_kernel<false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, CZ::LDS_BYTES));
    attr_set[dev][f32] = true;
  }
  {
    v3d::TimedScope ts("costreg_conv0", s);
    if (f32) conv0z_kernel<true><<<v3d::persistent_grid(tasks, 1), 512, CZ::LDS_BYTES, s>>>(p);
    else conv0z_kernel<false><<<v3d::persistent_grid(tasks, 1), 512, CZ::LDS_BYTES, s>>>(p);
  }
  V3D_CHECK_LAUNCH("conv0z_kernel");
  return V3D_OK;
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_conv0z_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  static unsigned long long h[8 * 1024];
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cz_phase), sizeof(h)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < 1024; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif
