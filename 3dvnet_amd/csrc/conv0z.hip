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

#ifndef V3D_CZ_ABLATE
#define V3D_CZ_ABLATE 0      // developer ablations: 1 no MFMAs, 2 no DMA, 3 no reduction / finalize
#endif

__global__ __launch_bounds__(256, 1) void conv0z_kernel(CZParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = input channel chunk
  const int kq = lane >> 4, jn = lane & 15;
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned ring_lds = smem_lds + (unsigned)wave * CZ::RING_BYTES;      // what M0 carries
  unsigned char* const ring = smem + wave * CZ::RING_BYTES;
  f32x4* const red = reinterpret_cast<f32x4*>(smem + 4 * CZ::RING_BYTES);

  // this wave's weight fragments: resident for the whole kernel
  bf16x8 a_hi[9], a_lo[9];
  {
    const u32x4* wq = reinterpret_cast<const u32x4*>(p.wp) + (size_t)wave * (9 * 2 * 64) + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      a_hi[k] = __builtin_bit_cast(bf16x8, wq[(k * 2) * 64]);
      a_lo[k] = __builtin_bit_cast(bf16x8, wq[(k * 2 + 1) * 64]);
    }
  }
  // B operand of block b, column jn, k group kq: slot (y, 2 xp + kq) of the tile row y + ky (ky: immediate offset)
  unsigned boff[CZ::NBLK];
#pragma unroll
  for (int b = 0; b < CZ::NBLK; ++b) {
    const int q = 16 * b + jn, y = q / CZ::NP, xp = q % CZ::NP;
    boff[b] = (unsigned)((y * CZ::IWS + 2 * xp + kq) * 16);
  }
  float sbias[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) sbias[r] = p.bias[r];
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = cz_lane_select(kq & 1, sbias[4 + r], sbias[r]);

  const size_t HW = (size_t)p.H * p.W, DHW = (size_t)p.D * HW;
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n_tasks);
  f32x4 acc[3][CZ::NBLK];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int b = 0; b < CZ::NBLK; ++b) acc[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int t = walk.t; t < walk.end; t += walk.step) {
    // task = (view, z segment, y tile, x tile), x fastest
    int tt = t;
    const int tx = tt % p.ntx; tt /= p.ntx;
    const int ty = tt % p.nty; tt /= p.nty;
    const int seg = tt % p.nseg;
    const int n = tt / p.nseg;
    const int oy0 = ty * CZ::TH, ox0 = tx * CZ::TW;
    const int z0 = seg * p.seg_len, z1 = min(z0 + p.seg_len, p.D);
    const int nsteps = z1 - z0 + 2;                           // input planes z0 - 1 .. z1

    // DMA role: lane = (row of the piece, slot of the row); slots 30, 31 are never read and always load (the plane's first
    // slot), so every piece is issued whatever the tile and the vmcnt arithmetic is exact
    unsigned voff[CZ::NPIECE];
    unsigned long long vmask[CZ::NPIECE];
    {
      const int j = lane >> 5, col = lane & 31;
      const int gx = ox0 - 1 + col;
#pragma unroll
      for (int i = 0; i < CZ::NPIECE; ++i) {
        const int gy = oy0 - 1 + 2 * i + j;
        const bool ok = col < CZ::TW + 2 && gx >= 0 && gx < p.W && gy >= 0 && gy < p.H;
        voff[i] = ok ? (unsigned)((gy * p.W + gx) * 16) : 0u;
        vmask[i] = __ballot(ok || col >= CZ::TW + 2);
      }
    }
    // out-of-volume slots of the tile are never written by the DMA: they are zeros from here on
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the previous task's copies have all been consumed)
      u32x4* const rz = reinterpret_cast<u32x4*>(ring);
#pragma unroll
      for (int i = 0; i < CZ::RING_BYTES / 1024; ++i) rz[i * 64 + lane] = (u32x4){0u, 0u, 0u, 0u};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // finalize role: this wave finishes blocks w and w + 4; lane (kq, jn) holds channels 4 (kq & 1) .. + 3 of the voxel
    // (y, 2 xp + (kq >> 1)) -- one 8-byte half of its hi slot and of its lo slot
    int fsp[2];
    bool fok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int b = wave + 4 * k;
      const int q = 16 * b + jn, y = q / CZ::NP, xp = q % CZ::NP;
      const int gy = oy0 + y, gx = ox0 + 2 * xp + (kq >> 1);
      fok[k] = b < CZ::NBLK && gy < p.H && gx < p.W;
      fsp[k] = gy * p.W + gx;
    }
    const char* const in_c = reinterpret_cast<const char*>(p.in) + ((size_t)(n * 4 + wave) * 2) * DHW * 16;
    u32x2* const outs = reinterpret_cast<u32x2*>(p.out) + ((size_t)n * 2 * DHW) * 2 + (kq & 1);

    // one plane of this wave's chunk -> ring slot `rs`: 5 pieces of hi rows, 5 of lo rows
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

    // ---- one input plane: step s handles zi = z0 - 1 + s; U = s % 3 fixes the ring slot and the accumulator slots ----
    auto step = [&](int s, auto u_c) __attribute__((always_inline)) {
      constexpr int U = decltype(u_c)::value;
      constexpr int A0 = (U + 1) % 3, A1 = U, A2 = (U + 2) % 3;        // accumulators of out planes zi + 1, zi, zi - 1
      const int zi = z0 - 1 + s;
      const bool valid = zi >= 0 && zi < p.D;
      // planes of steps <= s + 1 have been requested: at most the newest one may still be in flight
      if (s + 1 < nsteps && zi + 1 < p.D) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (s + 2 < nsteps && zi + 2 < p.D) issue(zi + 2, A2);           // ring slot of step s + 2 = the one step s - 1 read
      if (valid) {
        const unsigned char* const rb = ring + U * CZ::PLANE_BYTES;
        if (V3D_CZ_ABLATE != 1) {
          // 21 (ky, block) items, each 2 ds_read_b128 -> 9 MFMAs; the B fragments run kPre items ahead of the MFMAs (one
          // wave per SIMD: nobody else covers an LDS round trip), the scheduler is pinned to that order
          constexpr int NI = 3 * CZ::NBLK, kPre = 2;
          bf16x8 bh_[NI], bl_[NI];
          auto load = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / CZ::NBLK, b = i % CZ::NBLK;
            bh_[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + boff[b] + ky * (CZ::IWS * 16)));
            bl_[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + boff[b] + ky * (CZ::IWS * 16) + CZ::HL_BYTES));
          };
          auto item = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, ky = i / CZ::NBLK, b = i % CZ::NBLK;
            if constexpr (i + kPre < NI) load(std::integral_constant<int, i + kPre>{});
            const bf16x8 b_hi = bh_[i], b_lo = bl_[i];
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
            if constexpr (i + kPre < NI) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);                                   // 9 MFMAs
          };
          load(std::integral_constant<int, 0>{});
          load(std::integral_constant<int, 1>{});
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * kPre, 0);
          cz_static_for<0, NI>(item);
        }
      } else {
#pragma unroll
        for (int b = 0; b < CZ::NBLK; ++b) acc[A0][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (s < 2 || V3D_CZ_ABLATE == 3) return;
      // out plane zo = zi - 1 is complete in this wave's chunk: the four partial sums meet in LDS
      const int zo = zi - 1;
#pragma unroll
      for (int b = 0; b < CZ::NBLK; ++b) red[(wave * CZ::NBLK + b) * 64 + lane] = acc[A2][b];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int b = wave + 4 * k;
        if (b >= CZ::NBLK) continue;                               // wave-uniform
        f32x4 v = red[(0 * CZ::NBLK + b) * 64 + lane];
        v += red[(1 * CZ::NBLK + b) * 64 + lane];
        v += red[(2 * CZ::NBLK + b) * 64 + lane];
        v += red[(3 * CZ::NBLK + b) * 64 + lane];
        unsigned h[4], l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float val = fmaxf(v[r] + bias[r], 0.f);
          h[r] = cz_bf16_rne(val);
          l[r] = cz_bf16_rne(val - __uint_as_float(h[r] << 16));
        }
        if (fok[k]) {
          const size_t sp = (size_t)zo * HW + fsp[k];
          outs[sp * 2] = (u32x2){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
          outs[(DHW + sp) * 2] = (u32x2){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
        }
      }
      __syncthreads();                                             // red is free for the next plane
    };

    // prologue: planes of steps 0 and 1
    if (z0 - 1 >= 0) issue(z0 - 1, 0);
    issue(z0, 1);
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 3) {
      step(s, std::integral_constant<int, 0>{});
      if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, 1>{});
      if (s + 2 < nsteps) step(s + 2, std::integral_constant<int, 2>{});
    }
  }
}

}  // namespace

// conv0 + folded BN + ReLU of a batch of split variance volumes -> split activation (the fused path's hand-off formats).
int v3d::launch_conv0z(const void* in_split, const float* wbf, const float* bias, void* out_split, int n, int D, int H, int W,
                       hipStream_t s) {
  V3D_REQUIRE((long long)D * H * W * 16 < (1ll << 32), V3D_ERR_BAD_SHAPE, "conv0: volume too large for 32-bit plane offsets");
  CZParams p;
  p.in = in_split; p.wp = wbf; p.bias = bias; p.out = out_split;
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
  static bool attr_set[64] = {false};
  if (dev < 64 && !attr_set[dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv0z_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CZ::LDS_BYTES));
    attr_set[dev] = true;
  }
  {
    v3d::TimedScope ts("costreg_conv0", s);
    conv0z_kernel<<<v3d::persistent_grid(tasks, 1), 256, CZ::LDS_BYTES, s>>>(p);
  }
  V3D_CHECK_LAUNCH("conv0z_kernel");
  return V3D_OK;
}
