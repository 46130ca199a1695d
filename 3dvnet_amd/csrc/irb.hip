// an inverted-residual block of the MnasNet trunk (mv3d/subnetworks/mvsnet.py:55-73; torchvision
// _InvertedResidual: 1x1 expand + BN + ReLU -> k x k depthwise (stride s) + BN + ReLU -> 1x1 project + BN (+ identity shortcut)) as ONE
// kernel.  The three-launch path of round 5 (csrc/backbone.hip) writes and re-reads the expanded tensor twice -- 3 to 6 times the
// block's input -- and is what the trunk's 2.3 ms per 71 images of 256 x 320 were made of.  Here a workgroup owns a tile of 8 x TW
// output positions of one image and walks the expanded channels in SLICES OF 32:
//
//   X  expand:   E[region position, 32 ch] = ReLU(x[position, 0:cin] We[0:cin, slice] + be)    matrix cores, split-bf16 operands
//   W  depthwise D[tile position, 32 ch]   = ReLU(sum_taps E[position * s + tap] wd[tap] + bd)  fp32 VALU out of LDS
//   P  project:  acc[tile position, cout] += D[position, slice] Wp[slice, 0:cout]               matrix cores, accumulators live
//                                                                                                in registers across the slices
//
// so the expanded tensor exists only as one 32-channel slice of the tile's input region in LDS ("E", fp32: the depthwise taps are
// exact fp32) and as the split-bf16 A operand of the projection ("D").  Details:
//   * the tile's input region is (7 s + k) x ((TW - 1) s + k) positions; only its IN-IMAGE positions are expanded (row list
//     `rowofs`: region slot of the v-th valid position) -- the zero padding of the depthwise convolution is E's untouched zeros, and
//     the low-resolution maps (a 8 x 10 image is one tile whose 12 x 14 region is mostly padding) cost no matrix work for it;
//   * the input rows of a wave's region row blocks are loaded and split ONCE: they stay in registers as A fragments for all slices;
//   * the slice's weight images (expand fragments + bias, depthwise taps + bias, project fragments) are fetched one slice ahead
//     into registers and parked in LDS behind the barrier that frees their buffer; two barriers per slice;
//   * measured and withdrawn: the expansion with matrix rows = channels (a lane then holds four consecutive channels of its position:
//     four 16-byte E stores behind one address instead of sixteen 4-byte stores, 60 instead of 100 vector instructions per row block
//     and slice, E slots padded to 144 bytes against the 16-lanes-on-4-banks conflict): the 1/2-resolution blocks and the stem lost
//     10-20 % (10 more registers and 1.6 KB more LDS per workgroup: four instead of five workgroups per CU), the others gained 5 %;
//     8 x 16 tiles for the stride-1 blocks at 1/4 and 1/8 resolution and the stem (half the tiles, balanced projection tasks, 49 KB of
//     LDS = three instead of five workgroups per CU): blocks 0.98 -> 1.01 ms, stem 0.169 -> 0.164;
//   * LDS reads that feed vector instructions are 8 bytes per lane (16-byte reads beside matrix instructions in
//     flight have returned stale lanes on this chip); the 16-byte reads here all feed matrix instructions.
// Arithmetic: split-bf16 matrix operands (hi*hi + hi*lo + lo*hi, fp32 accumulation) as everywhere on this path; the exact-fp32
// variant of the block is the three-launch path.
#include <cstring>
#include <utility>
#include <vector>

#include "v3d_common.h"

struct v3d_irb_weights {
  int cin, mid, cout, ks, stride, residual, stem;
  int ns, csteps, ncbo;              // slices of 32 expanded channels, 16-channel K steps of the expansion, 32-channel blocks of cout
  char* dev;
  size_t xw_ofs, dw_ofs, pw_ofs, bp_ofs;
  size_t xw_slice, dw_slice, pw_slice;
};

int v3d_irb_launch(const v3d_irb_weights* h, const float* x, int n, int H, int W, int ih, int iw, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct IrbParams {
  const float* x;
  float* out;
  const char* xw;        // [ns][csteps][hi, lo][64 lanes][16 B] + expand bias [32] per slice
  const char* dw;        // [ns][k * k taps + bias][32] fp32
  const char* pw;        // [ns][ncbo][2 steps][hi, lo][64 lanes][16 B]
  const float* bp;       // [ncbo * 32]
  int n, H, W, Ho, Wo, cin, cout, ns, residual, tiles_x, tiles_y;
  int ih, iw;            // STEM: the image's sides (x = image [n, 3, ih, iw], H x W = the stem convolution's output)
  int nsg;               // slice groups: workgroup (tile, group) walks ns / nsg slices and leaves a partial sum in `part`
  float* part;           // [nsg][n * Ho * Wo][cout] when nsg > 1
};

__device__ __forceinline__ unsigned irb_pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2_));
}
// x = hi + lo (hi = RNE_bf16(x), lo = RNE_bf16(x - hi)), two values per word
__device__ __forceinline__ void irb_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = irb_pack_bf16x2(a, b);
  lo = irb_pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// 8-byte LDS read by hand: the compiler merges neighbouring 8-byte reads into 16-byte-per-lane instructions (ds_read_b128 /
// ds_read2_b64), which must not feed vector instructions beside matrix instructions in flight, and `volatile`
// turns them into waited-for flat loads.  The result is valid behind irb_lds_wait (the wait the compiler cannot count for us);
// the compiler's own LDS traffic only ever waits longer because of these reads.
template <int OFS>
__device__ __forceinline__ f32x2 irb_lds_read8(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFS));
  return v;
}
__device__ __forceinline__ void irb_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// (consumers of v are ordered behind the wait: the empty statement redefines v after it)
__device__ __forceinline__ void irb_tie(f32x2& v) { asm volatile("" : "+v"(v)); }
template <int... Is, class F>
__device__ __forceinline__ void irb_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
__device__ __forceinline__ unsigned irb_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

template <int KS, int S, int NCBO, int TW, int CSTEPS, int RBW>
struct IrbCfg {
  static constexpr int TH = 8, T = TH * TW, RBP = (T + 31) / 32;            // tile positions, their row blocks of 32
  static constexpr int RH = (TH - 1) * S + KS, RW = (TW - 1) * S + KS, R = RH * RW;
  static constexpr int NRB = 4 * RBW;                                         // region row blocks the four waves can hold
  static constexpr int NT = (RBP * NCBO + 3) / 4;                             // project tasks (row block, column block) per wave
  static constexpr int XW_BYTES = CSTEPS * 2048 + 128, DW_BYTES = (KS * KS + 1) * 128, PW_BYTES = NCBO * 4096;
  static constexpr int NXP = (XW_BYTES / 16 + 255) / 256, NDP = (DW_BYTES / 16 + 255) / 256;
  // LDS map (bytes)
  static constexpr int E_OFS = 0, E_BYTES = (R + 1) * 128;                    // [region slot][32] fp32; slot R = dump row
  // D: eight planes [step][hi, lo][k half] of [row][16 B]; a plane is 32 bytes longer than its rows so that the planes start 8 banks
  // apart (the depthwise role stores 4 bytes per lane into four planes at once: with plane starts 1 KB apart they were 8 lanes per bank;
  // time unchanged: the LDS is not what the kernel waits for)
  static constexpr int DP = RBP * 32 * 16 + 32;
  static constexpr int D_OFS = E_OFS + E_BYTES, D_BYTES = 8 * DP;
  static constexpr int XW_OFS = D_OFS + D_BYTES;
  static constexpr int PW_OFS = XW_OFS + XW_BYTES;
  static constexpr int DW_OFS = PW_OFS + PW_BYTES;
  static constexpr int RO_OFS = DW_OFS + DW_BYTES;                            // rowofs [NRB * 32] u16
  static constexpr int LDS = RO_OFS + NRB * 32 * 2;
  // (the host checks that every tile's IN-IMAGE region positions fit the NRB row blocks: irb_pick)
  static_assert(LDS <= 160 * 1024, "LDS of one CU");
  static_assert(DW_BYTES / 16 <= 256, "one 16-byte piece of the depthwise image per thread");
};

#ifdef V3D_IRB_PHASE
// developer build only (scripts/micro/irb_check.py): cycles of wave 0 per workgroup, summed: 0 prologue, 1 X, 2 first barrier + image
// parking, 3 W, 4 second barrier, 5 P, 6 epilogue, 7 workgroups
__device__ unsigned long long g_irb_phase[8];
#define IRB_PH_DECL long long ph_t = __builtin_readcyclecounter(); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define IRB_PH(i) do { long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#define IRB_PH_FLUSH do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_irb_phase[i_], (unsigned long long)ph_acc[i_]); atomicAdd(&g_irb_phase[7], 1ull); } } while (0)
#else
#define IRB_PH_DECL
#define IRB_PH(i)
#define IRB_PH_FLUSH
#endif

// OCC: waves per SIMD the register allocation must leave room for (= workgroups per CU; the high-resolution blocks have many small
// tiles and hide each other's barriers and LDS round trips, the low-resolution ones have one tile per CU at most)
// STEM: the block is the trunk's first three layers (mvsnet.py:60, torchvision mnasnet layers 0-7): the "expansion" is the 3x3 /
// stride 2 / pad 1 convolution 3 -> 32 of the NCHW image -- its 27 (channel, ky, kx) taps gathered as the K dimension of the same
// matrix product -- followed by the 3x3 depthwise and the 1x1 projection to 16 channels.
template <int KS, int S, int NCBO, int TW, int CSTEPS, int RBW, int OCC, bool STEM = false>
__global__ __launch_bounds__(256, OCC) void irb_kernel(IrbParams p) {
  using C = IrbCfg<KS, S, NCBO, TW, CSTEPS, RBW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const E = reinterpret_cast<float*>(smem + C::E_OFS);
  unsigned char* const D = smem + C::D_OFS;
  unsigned char* const XW = smem + C::XW_OFS;
  unsigned char* const PW = smem + C::PW_OFS;
  unsigned char* const DWL = smem + C::DW_OFS;
  unsigned short* const rowofs = reinterpret_cast<unsigned short*>(smem + C::RO_OFS);

  IRB_PH_DECL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, i = lane & 31;
  int b = v3d::xcd_contiguous_block();
  const int sg = b % p.nsg;                               // (the groups of a tile are neighbours: they read the same input rows)
  b /= p.nsg;
  const int s_begin = (p.ns * sg) / p.nsg, s_end = (p.ns * (sg + 1)) / p.nsg;
  const int tx0 = (b % p.tiles_x) * TW;
  b /= p.tiles_x;
  const int ty0 = (b % p.tiles_y) * C::TH;
  const int img = b / p.tiles_y;
  // the tile's input region and its in-image part (never empty: the tile's first output position reads its own centre tap)
  const int ry0 = ty0 * S - KS / 2, rx0 = tx0 * S - KS / 2;
  const int vy0 = max(ry0, 0), vx0 = max(rx0, 0);
  const int vh = min(ry0 + C::RH, p.H) - vy0, vw = min(rx0 + C::RW, p.W) - vx0;
  const int nv = vh * vw;

  // every global read of the prologue first (one round trip): the input rows of this wave's region row blocks wave, wave + 4, ...
  // -- lane (g, i) = channels 16 st + 8 g .. + 7 of valid position 32 rb + i; rows behind the last valid position repeat it and land
  // in E's dump row -- and the first slice's weight images
  f32x4 araw[RBW][CSTEPS][2];
#pragma unroll
  for (int q = 0; q < RBW; ++q) {
    const int v = min((wave + 4 * q) * 32 + i, nv - 1);
    const int vy = vy0 + v / vw, vx = vx0 + v % vw;
    if constexpr (STEM) {
      // k = 16 st + 8 g + e = 9 c + 3 ky + kx: pixel (2 vy + ky - 1, 2 vx + kx - 1) of channel c, zero outside the image / for k >= 27
      const float* const im = p.x + (size_t)img * 3 * p.ih * p.iw;
#pragma unroll
      for (int st = 0; st < CSTEPS; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ka = 16 * st + e, kb = ka + 8;                        // the lane's k for g = 0 / 1
          const int c = g ? kb / 9 : ka / 9, ky = g ? (kb % 9) / 3 : (ka % 9) / 3, kx = g ? kb % 3 : ka % 3;
          const int iy = 2 * vy + ky - 1, ix = 2 * vx + kx - 1;
          const bool ok = (g ? kb : ka) < 27 && (unsigned)iy < (unsigned)p.ih && (unsigned)ix < (unsigned)p.iw;
          const float v = im[((size_t)(ok ? c : 0) * p.ih + (ok ? iy : 0)) * p.iw + (ok ? ix : 0)];
          araw[q][st][e >> 2][e & 3] = ok ? v : 0.f;
        }
    } else {
    const float* const xr = p.x + ((size_t)(img * p.H + vy) * p.W + vx) * p.cin;
#pragma unroll
    for (int st = 0; st < CSTEPS; ++st) {
      const int k0 = 16 * st + 8 * g;
      const int ko = k0 < p.cin ? k0 : 0;              // (cin is a multiple of 8: the eight channels are inside or outside together)
      araw[q][st][0] = *reinterpret_cast<const f32x4*>(xr + ko);
      araw[q][st][1] = *reinterpret_cast<const f32x4*>(xr + ko + 4);
    }
    }
  }
  {
    u32x4 xw0[C::NXP], pw0[NCBO], dw0;
#pragma unroll
    for (int k = 0; k < C::NXP; ++k)
      xw0[k] = reinterpret_cast<const u32x4*>(p.xw + (size_t)s_begin * C::XW_BYTES)[min(tid + 256 * k, C::XW_BYTES / 16 - 1)];
#pragma unroll
    for (int k = 0; k < NCBO; ++k) pw0[k] = reinterpret_cast<const u32x4*>(p.pw + (size_t)s_begin * C::PW_BYTES)[tid + 256 * k];
    dw0 = reinterpret_cast<const u32x4*>(p.dw + (size_t)s_begin * C::DW_BYTES)[min(tid, C::DW_BYTES / 16 - 1)];
    // ... and the LDS set-up while they travel
    for (int k = tid; k < C::E_BYTES / 16; k += 256) reinterpret_cast<u32x4*>(E)[k] = (u32x4){0u, 0u, 0u, 0u};
    for (int k = tid; k < C::D_BYTES / 16; k += 256) reinterpret_cast<u32x4*>(D)[k] = (u32x4){0u, 0u, 0u, 0u};
    for (int v = tid; v < C::NRB * 32; v += 256)
      rowofs[v] = (unsigned short)(v < nv ? (vy0 - ry0 + v / vw) * C::RW + (vx0 - rx0 + v % vw) : C::R);
#pragma unroll
    for (int k = 0; k < C::NXP; ++k)
      if (tid + 256 * k < C::XW_BYTES / 16) reinterpret_cast<u32x4*>(XW)[tid + 256 * k] = xw0[k];
#pragma unroll
    for (int k = 0; k < NCBO; ++k) reinterpret_cast<u32x4*>(PW)[tid + 256 * k] = pw0[k];
    if (tid < C::DW_BYTES / 16) reinterpret_cast<u32x4*>(DWL)[tid] = dw0;
  }
  u32x4 ah[RBW][CSTEPS], al[RBW][CSTEPS];
#pragma unroll
  for (int q = 0; q < RBW; ++q)
#pragma unroll
    for (int st = 0; st < CSTEPS; ++st) {
      const f32x4 a0 = araw[q][st][0], a1 = araw[q][st][1];
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      irb_split2(a0.x, a0.y, h0, l0);
      irb_split2(a0.z, a0.w, h1, l1);
      irb_split2(a1.x, a1.y, h2, l2);
      irb_split2(a1.z, a1.w, h3, l3);
      const unsigned keep = 16 * st + 8 * g < p.cin ? 0xffffffffu : 0u;
      ah[q][st] = (u32x4){h0, h1, h2, h3} & (u32x4){keep, keep, keep, keep};
      al[q][st] = (u32x4){l0, l1, l2, l3} & (u32x4){keep, keep, keep, keep};
    }
  f32x16 pacc[C::NT];
#pragma unroll
  for (int q = 0; q < C::NT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[q][r] = 0.f;
  __syncthreads();

  IRB_PH(0);
  const unsigned e_lds = irb_lds_addr(E), dw_lds = irb_lds_addr(DWL), ro_lds = irb_lds_addr(rowofs);
  u32x4 pw_reg[NCBO];                                     // the NEXT slice's project image, a whole slice in flight
#pragma unroll 1
  for (int s = s_begin; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    // (1) the next slice's images on their way.  Expand + depthwise (s + 1) are parked in LDS behind this slice's W; the project
    // image (s + 1) behind the NEXT slice's first barrier, when every wave is done with P (s).
    u32x4 xw_reg[C::NXP], dw_reg;
    {
      const int sn = more ? s + 1 : s;
      const u32x4* const xs = reinterpret_cast<const u32x4*>(p.xw + (size_t)sn * C::XW_BYTES);
#pragma unroll
      for (int k = 0; k < C::NXP; ++k) xw_reg[k] = xs[min(tid + 256 * k, C::XW_BYTES / 16 - 1)];
      dw_reg = reinterpret_cast<const u32x4*>(p.dw + (size_t)sn * C::DW_BYTES)[min(tid, C::DW_BYTES / 16 - 1)];
    }
    // (2) X: this wave's row blocks x the slice's 32 channels
    {
      const float be = reinterpret_cast<const float*>(XW + CSTEPS * 2048)[i];
#pragma unroll
      for (int q = 0; q < RBW; ++q) {
        const int rb = wave + 4 * q;
        if (rb * 32 >= nv) continue;                       // (wave-uniform)
        // (two accumulators when the K loop is long: a chain of dependent matrix instructions pays each one's full latency)
        constexpr int NA = CSTEPS >= 4 ? 2 : 1;
        f32x16 accs[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[a][r] = 0.f;
#pragma unroll
        for (int st = 0; st < CSTEPS; ++st) {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(XW + (st * 2 + 0) * 1024)[lane]);
          const bf16x8 wl = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(XW + (st * 2 + 1) * 1024)[lane]);
          f32x16& a = accs[st % NA];
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[q][st]), wh, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[q][st]), wl, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[q][st]), wh, a, 0, 0, 0);
        }
        f32x16 acc = accs[0];
        if (NA == 2) acc += accs[1];
        // lane (g, n = i) holds channel n of the valid positions 32 rb + 8 j + 4 g + r: their region slots, four u16 per read
        const unsigned ra = ro_lds + (unsigned)(rb * 32 + 4 * g) * 2u;
        f32x2 ro[4] = {irb_lds_read8<0>(ra), irb_lds_read8<16>(ra), irb_lds_read8<32>(ra), irb_lds_read8<48>(ra)};
        irb_lds_wait();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          irb_tie(ro[j]);
          const unsigned w0 = __float_as_uint(ro[j].x), w1 = __float_as_uint(ro[j].y);
          E[(w0 & 0xffffu) * 32 + i] = fmaxf(acc[4 * j + 0] + be, 0.f);
          E[(w0 >> 16) * 32 + i] = fmaxf(acc[4 * j + 1] + be, 0.f);
          E[(w1 & 0xffffu) * 32 + i] = fmaxf(acc[4 * j + 2] + be, 0.f);
          E[(w1 >> 16) * 32 + i] = fmaxf(acc[4 * j + 3] + be, 0.f);
        }
      }
    }
    IRB_PH(1);
    __syncthreads();
    // (4) every wave is done with P (s - 1): the project image of this slice (requested a slice ago) takes its buffer
    if (s > s_begin) {
#pragma unroll
      for (int k = 0; k < NCBO; ++k) reinterpret_cast<u32x4*>(PW)[tid + 256 * k] = pw_reg[k];
    }
    if (more) {
      const u32x4* const ps = reinterpret_cast<const u32x4*>(p.pw + (size_t)(s + 1) * C::PW_BYTES);
#pragma unroll
      for (int k = 0; k < NCBO; ++k) pw_reg[k] = ps[tid + 256 * k];
    }
    IRB_PH(2);
    // (5) W: depthwise taps out of E; a thread = (two channels, half a tile row = NX neighbouring outputs): a tap row is NIN + KS
    // 8-byte reads for NX x KS packed FMAs, the 16 channel pairs of a position are one 128-byte row: every bank once per 16 lanes
    {
      constexpr int NX = TW / 2, NIN = (NX - 1) * S + KS;
      static_assert(C::TH * 2 * 16 == 256, "one (half row, channel pair) per thread");
      const int cp = tid & 15, hr = tid >> 4, ty = hr >> 1, txh = (hr & 1) * NX;
      unsigned ea = e_lds + (unsigned)((ty * S) * C::RW + txh * S) * 128u + (unsigned)cp * 8u, wa = dw_lds + (unsigned)cp * 8u;
      f32x2 bias = irb_lds_read8<KS * KS * 128>(wa);
      irb_lds_wait();
      irb_tie(bias);
      f32x2 acc[NX];
#pragma unroll
      for (int o = 0; o < NX; ++o) acc[o] = bias;
      // (one tap row at a time: unrolled, the scheduler issues every row's reads up front)
#pragma unroll 1
      for (int ky = 0; ky < KS; ++ky) {
        f32x2 w[KS], v[NIN];
        irb_static_for(std::make_integer_sequence<int, KS>{}, [&](auto kx) __attribute__((always_inline)) { w[kx.value] = irb_lds_read8<kx.value * 128>(wa); });
        irb_static_for(std::make_integer_sequence<int, NIN>{}, [&](auto jx) __attribute__((always_inline)) { v[jx.value] = irb_lds_read8<jx.value * 128>(ea); });
        irb_lds_wait();
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) irb_tie(w[kx]);
#pragma unroll
        for (int jx = 0; jx < NIN; ++jx) irb_tie(v[jx]);
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int o = 0; o < NX; ++o) acc[o] = __builtin_elementwise_fma(v[o * S + kx], w[kx], acc[o]);
        ea += C::RW * 128;
        wa += KS * 128;
      }
      // ReLU, split, -> the A-operand layout of the projection: channels 2 cp, 2 cp + 1 = bytes [4 (cp & 3), + 4) of the 16-byte chunk
      // (k half (cp >> 2) & 1) of step cp >> 3
      unsigned char* const d = D + (((cp >> 3) * 2 + 0) * 2 + ((cp >> 2) & 1)) * C::DP + (ty * TW + txh) * 16 + (cp & 3) * 4;
#pragma unroll
      for (int o = 0; o < NX; ++o) {
        const f32x2 r = __builtin_elementwise_max(acc[o], (f32x2){0.f, 0.f});
        unsigned hi, lo;
        irb_split2(r.x, r.y, hi, lo);
        *reinterpret_cast<unsigned*>(d + o * 16) = hi;
        *reinterpret_cast<unsigned*>(d + o * 16 + 2 * C::DP) = lo;
      }
    }
    IRB_PH(3);
    // X (s) finished before the first barrier: the expand image of the next slice takes its buffer (visible behind the next barrier)
    if (more) {
#pragma unroll
      for (int k = 0; k < C::NXP; ++k)
        if (tid + 256 * k < C::XW_BYTES / 16) reinterpret_cast<u32x4*>(XW)[tid + 256 * k] = xw_reg[k];
    }
    __syncthreads();
    IRB_PH(4);
    // (7) W (s) is done with the depthwise image
    if (more && tid < C::DW_BYTES / 16) reinterpret_cast<u32x4*>(DWL)[tid] = dw_reg;
    // (8) P: this wave's (row block, column block) tasks, two K steps of 16 expanded channels
#pragma unroll
    for (int q = 0; q < C::NT; ++q) {
      const int k = wave + 4 * q;
      if (k >= C::RBP * NCBO) continue;                    // (wave-uniform)
      const int rbp = k / NCBO, cb = k - rbp * NCBO;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 dh = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(D + ((st * 2 + 0) * 2 + g) * C::DP + rbp * 32 * 16)[i]);
        const bf16x8 dl = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(D + ((st * 2 + 1) * 2 + g) * C::DP + rbp * 32 * 16)[i]);
        const bf16x8 wh = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(PW + ((cb * 2 + st) * 2 + 0) * 1024)[lane]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(PW + ((cb * 2 + st) * 2 + 1) * 1024)[lane]);
        pacc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, wh, pacc[q], 0, 0, 0);
        pacc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh, wl, pacc[q], 0, 0, 0);
        pacc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl, wh, pacc[q], 0, 0, 0);
      }
    }
    IRB_PH(5);
  }

  // bias (+ shortcut) and store: lane (g, n = i) holds output channel 32 cb + n of the tile positions 32 rbp + 8 j + 4 g + r
#pragma unroll
  for (int q = 0; q < C::NT; ++q) {
    const int k = wave + 4 * q;
    if (k >= C::RBP * NCBO) continue;
    const int rbp = k / NCBO, cb = k - rbp * NCBO;
    const int co = cb * 32 + i;
    if (co >= p.cout) continue;
    const float bs = p.bp[co];
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) {
      const int t = rbp * 32 + 8 * (r16 >> 2) + 4 * g + (r16 & 3);
      const int ty = t / TW, tx = t - ty * TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      if (t >= C::T || oy >= p.Ho || ox >= p.Wo) continue;
      const size_t idx = ((size_t)(img * p.Ho + oy) * p.Wo + ox) * p.cout + co;
      if (p.nsg > 1) {                                     // partial sum of this slice group; irb_reduce_kernel finishes
        p.part[(size_t)sg * ((size_t)p.n * p.Ho * p.Wo * p.cout) + idx] = pacc[q][r16];
        continue;
      }
      float v = pacc[q][r16] + bs;
      if (p.residual) v += p.x[idx];
      p.out[idx] = v;
    }
  }
  IRB_PH(6);
  IRB_PH_FLUSH;
}

// out = sum of the slice groups' partial sums (in group order) + bias (+ x): one thread per 4 output channels
__global__ __launch_bounds__(256) void irb_reduce_kernel(const float* __restrict__ part, int nsg, size_t total, const float* __restrict__ bias,
                                                         const float* __restrict__ x, int cout, float* __restrict__ out) {
  const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (k >= total) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(part + k);
  for (int g = 1; g < nsg; ++g) v += *reinterpret_cast<const f32x4*>(part + (size_t)g * total + k);
  v += *reinterpret_cast<const f32x4*>(bias + k % (size_t)cout);
  if (x) v += *reinterpret_cast<const f32x4*>(x + k);
  *reinterpret_cast<f32x4*>(out + k) = v;
}

unsigned irb_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
float irb_bf16_value(unsigned h) {
  const unsigned u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// the instantiations the MnasNet-1.0 trunk needs at image sides that are multiples of 32 (and at 240 x 320); anything else takes
// the three-launch path
struct IrbVariant {
  int stem, ks, s, ncbo, tw, csteps, rbw;
  void (*kernel)(IrbParams);
  int lds;
};
#define V3D_IRB_VARIANT(KS_, S_, NCBO_, TW_, CSTEPS_, RBW_, OCC_) \
  {0, KS_, S_, NCBO_, TW_, CSTEPS_, RBW_, irb_kernel<KS_, S_, NCBO_, TW_, CSTEPS_, RBW_, OCC_>, IrbCfg<KS_, S_, NCBO_, TW_, CSTEPS_, RBW_>::LDS}
const IrbVariant kIrbVariants[] = {
    {1, 3, 1, 1, 8, 2, 1, irb_kernel<3, 1, 1, 8, 2, 1, 3, true>, IrbCfg<3, 1, 1, 8, 2, 1>::LDS},      // the stem
    V3D_IRB_VARIANT(3, 2, 1, 8, 1, 3, 2),   V3D_IRB_VARIANT(3, 1, 1, 8, 2, 1, 3),   V3D_IRB_VARIANT(5, 2, 2, 8, 2, 3, 2),
    V3D_IRB_VARIANT(5, 1, 2, 8, 3, 2, 2),   V3D_IRB_VARIANT(5, 2, 3, 10, 3, 4, 2),  V3D_IRB_VARIANT(5, 1, 3, 10, 5, 1, 2),
    V3D_IRB_VARIANT(3, 1, 3, 10, 5, 1, 2),  V3D_IRB_VARIANT(3, 1, 3, 10, 6, 1, 2),
    // the 1/32-resolution blocks: one tile per image; the expanded channels are shared out over slice groups (irb_groups)
    V3D_IRB_VARIANT(5, 2, 6, 10, 6, 3, 1),  V3D_IRB_VARIANT(5, 1, 6, 10, 12, 1, 1), V3D_IRB_VARIANT(3, 1, 10, 10, 12, 1, 1),
};
#undef V3D_IRB_VARIANT

// the largest number of in-image region positions of any tile
int irb_max_valid(int H, int W, int Ho, int Wo, int ks, int s, int tw) {
  const int rh = 7 * s + ks, rw = (tw - 1) * s + ks;
  int mh = 0, mw = 0;
  for (int ty0 = 0; ty0 < Ho; ty0 += 8) {
    const int y0 = ty0 * s - ks / 2, a = y0 < 0 ? 0 : y0, bnd = y0 + rh < H ? y0 + rh : H;
    if (bnd - a > mh) mh = bnd - a;
  }
  for (int tx0 = 0; tx0 < Wo; tx0 += tw) {
    const int x0 = tx0 * s - ks / 2, a = x0 < 0 ? 0 : x0, bnd = x0 + rw < W ? x0 + rw : W;
    if (bnd - a > mw) mw = bnd - a;
  }
  return mh * mw;
}

const IrbVariant* irb_pick(const v3d_irb_weights* h, int H, int W) {
  const int Ho = (H + h->stride - 1) / h->stride, Wo = (W + h->stride - 1) / h->stride;
  // tiles of 8 x 10 where they tile the map exactly (the 16 x 20 and 8 x 10 maps of 256 x 320 images), else 8 x 8
  const int pref = (Wo % 10 == 0 && Wo <= 20) ? 10 : 8;
  for (int tw : {pref, 18 - pref})
    for (const IrbVariant& v : kIrbVariants)
      if (v.stem == h->stem && v.ks == h->ks && v.s == h->stride && v.ncbo == h->ncbo && v.csteps == h->csteps && v.tw == tw &&
          irb_max_valid(H, W, Ho, Wo, h->ks, h->stride, tw) <= 4 * v.rbw * 32)
        return &v;
  return nullptr;
}

}  // namespace

// HOST weights with eval-mode BatchNorm folded: w_expand [mid, cin], w_dw [mid, k, k], w_project [cout, mid], biases [mid] / [mid] / [cout]
namespace {
int irb_pack_impl(const float* w_expand, const float* b_expand, const float* w_dw, const float* b_dw, const float* w_project,
                  const float* b_project, int cin, int mid, int cout, int ksize, int stride, int residual, int stem,
                  v3d_irb_weights** out_handle) {
  v3d_irb_weights* h = new v3d_irb_weights();
  h->cin = cin; h->mid = mid; h->cout = cout; h->ks = ksize; h->stride = stride; h->residual = residual; h->stem = stem;
  h->ns = (mid + 31) / 32; h->csteps = (cin + 15) / 16; h->ncbo = (cout + 31) / 32;
  h->xw_slice = (size_t)h->csteps * 2048 + 128;
  h->dw_slice = (size_t)(ksize * ksize + 1) * 128;
  h->pw_slice = (size_t)h->ncbo * 4096;
  h->xw_ofs = 0;
  h->dw_ofs = h->xw_ofs + h->xw_slice * h->ns;
  h->pw_ofs = h->dw_ofs + h->dw_slice * h->ns;
  h->bp_ofs = h->pw_ofs + h->pw_slice * h->ns;
  std::vector<unsigned char> host(h->bp_ofs + (size_t)h->ncbo * 32 * 4, 0);
  auto put_bf16 = [&](size_t byte_ofs_hi, size_t byte_ofs_lo, float w) {
    const unsigned hi = irb_rne(w), lo = irb_rne(w - irb_bf16_value(hi));
    const unsigned short h16 = (unsigned short)hi, l16 = (unsigned short)lo;
    memcpy(&host[byte_ofs_hi], &h16, 2);
    memcpy(&host[byte_ofs_lo], &l16, 2);
  };
  for (int s = 0; s < h->ns; ++s) {
    // expand fragments: lane (g, n): expanded channel 32 s + n, input channels 16 st + 8 g + e
    for (int st = 0; st < h->csteps; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int m = 32 * s + (lane & 31), k = 16 * st + 8 * (lane >> 5) + e;
          const float w = (m < mid && k < cin) ? w_expand[(size_t)m * cin + k] : 0.f;
          const size_t base = h->xw_ofs + s * h->xw_slice + (size_t)st * 2048 + lane * 16 + e * 2;
          put_bf16(base, base + 1024, w);
        }
    float* const be = reinterpret_cast<float*>(&host[h->xw_ofs + s * h->xw_slice + (size_t)h->csteps * 2048]);
    float* const dwp = reinterpret_cast<float*>(&host[h->dw_ofs + s * h->dw_slice]);
    for (int n = 0; n < 32; ++n) {
      const int m = 32 * s + n;
      be[n] = m < mid ? b_expand[m] : 0.f;
      for (int t = 0; t < ksize * ksize; ++t) dwp[t * 32 + n] = m < mid ? w_dw[(size_t)m * ksize * ksize + t] : 0.f;
      dwp[ksize * ksize * 32 + n] = m < mid ? b_dw[m] : 0.f;
    }
    // project fragments: lane (g, n): output channel 32 cb + n, expanded channels 32 s + 16 st + 8 g + e
    for (int cb = 0; cb < h->ncbo; ++cb)
      for (int st = 0; st < 2; ++st)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int co = 32 * cb + (lane & 31), m = 32 * s + 16 * st + 8 * (lane >> 5) + e;
            const float w = (co < cout && m < mid) ? w_project[(size_t)co * mid + m] : 0.f;
            const size_t base = h->pw_ofs + s * h->pw_slice + ((size_t)(cb * 2 + st) * 2) * 1024 + lane * 16 + e * 2;
            put_bf16(base, base + 1024, w);
          }
  }
  float* const bp = reinterpret_cast<float*>(&host[h->bp_ofs]);
  for (int co = 0; co < cout; ++co) bp[co] = b_project[co];
  hipError_t e = hipMalloc((void**)&h->dev, host.size());
  if (e != hipSuccess) { delete h; return v3d::fail(V3D_ERR_HIP, "hipMalloc(block weights): %s", hipGetErrorString(e)); }
  e = hipMemcpy(h->dev, host.data(), host.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return v3d::fail(V3D_ERR_HIP, "hipMemcpy(block weights): %s", hipGetErrorString(e)); }
  *out_handle = h;
  return V3D_OK;
}
}  // namespace

extern "C" int v3d_irb_pack(const float* w_expand, const float* b_expand, const float* w_dw, const float* b_dw, const float* w_project,
                            const float* b_project, int cin, int mid, int cout, int ksize, int stride, int residual,
                            v3d_irb_weights** out_handle) {
  V3D_REQUIRE(w_expand && b_expand && w_dw && b_dw && w_project && b_project && out_handle, V3D_ERR_BAD_ARG, "v3d_irb_pack: null argument");
  V3D_REQUIRE(cin >= 8 && cin % 8 == 0 && mid >= 8 && mid % 8 == 0 && cout >= 8 && cout % 8 == 0 && (ksize == 3 || ksize == 5) &&
                  (stride == 1 || stride == 2) && (!residual || (cin == cout && stride == 1)),
              V3D_ERR_BAD_SHAPE, "v3d_irb_pack: cin=%d mid=%d cout=%d k=%d stride=%d residual=%d", cin, mid, cout, ksize, stride, residual);
  return irb_pack_impl(w_expand, b_expand, w_dw, b_dw, w_project, b_project, cin, mid, cout, ksize, stride, residual, 0, out_handle);
}

// The trunk's first three layers as one block: w_stem [32, 3, 3, 3] (= [32, 27], (channel, ky, kx) order), w_dw [32, 3, 3], w_pw [16, 32],
// BatchNorm folded, biases [32] / [32] / [16]
extern "C" int v3d_stem_block_pack(const float* w_stem, const float* b_stem, const float* w_dw, const float* b_dw, const float* w_pw,
                                   const float* b_pw, v3d_irb_weights** out_handle) {
  V3D_REQUIRE(w_stem && b_stem && w_dw && b_dw && w_pw && b_pw && out_handle, V3D_ERR_BAD_ARG, "v3d_stem_block_pack: null argument");
  std::vector<float> we(32 * 32, 0.f);                   // K padded from 27 to 32 with zero weights
  for (int m = 0; m < 32; ++m)
    for (int k = 0; k < 27; ++k) we[m * 32 + k] = w_stem[m * 27 + k];
  return irb_pack_impl(we.data(), b_stem, w_dw, b_dw, w_pw, b_pw, 32, 32, 16, 3, 1, 0, 1, out_handle);
}

// image [n, 3, IH, IW] (NCHW, even sides) -> out [n, IH / 2, IW / 2, 16] channels-last
extern "C" int v3d_stem_block_f32(const v3d_irb_weights* h, const float* image, int n, int IH, int IW, float* out, void* stream) {
  V3D_REQUIRE(h && h->stem && image && out, V3D_ERR_BAD_ARG, "v3d_stem_block_f32: null argument or not a stem handle");
  V3D_REQUIRE(n >= 0 && IH >= 2 && IW >= 2 && IH % 2 == 0 && IW % 2 == 0, V3D_ERR_BAD_SHAPE, "v3d_stem_block_f32: n=%d image %d x %d (even sides)", n, IH, IW);
  if (n == 0) return V3D_OK;
  return v3d_irb_launch(h, image, n, IH / 2, IW / 2, IH, IW, out, nullptr, 0, stream);
}

extern "C" void v3d_irb_free(v3d_irb_weights* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  delete h;
}

#ifdef V3D_IRB_PHASE
extern "C" int v3d_debug_irb_phase(unsigned long long* out8, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_irb_phase), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_irb_phase), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

namespace {
// Slice groups per tile: a map with fewer tiles than the chip has CUs (71 images of 8 x 10 = 71 tiles) hands the slices of a tile to
// several workgroups, as many as keep every workgroup on its own CU; their partial sums meet in the workspace.  (Aiming at two or
// three workgroups per CU -- which would also split the 284 tiles of the 1/16-resolution maps -- measured 0.98 -> 1.05 / 1.10 ms
// for the trunk's 16 blocks: the prologue and the reducing launch cost more than the tail they remove.)
int irb_groups(const v3d_irb_weights* h, long long tiles) {
  int n_cu = (int)v3d::persistent_grid(1 << 20, 1);
  long long g = n_cu / (tiles > 0 ? tiles : 1);
  if (g > h->ns / 4) g = h->ns / 4;                       // at least four slices per group: the prologue is paid per group
  return g < 1 ? 1 : (int)g;
}
}  // namespace

// bytes of workspace v3d_irb_nhwc_f32 needs for n images of H x W (0 for most maps)
extern "C" size_t v3d_irb_workspace_bytes(const v3d_irb_weights* h, int n, int H, int W) {
  if (!h || n <= 0 || H <= 0 || W <= 0) return 0;
  const IrbVariant* v = irb_pick(h, H, W);
  if (!v) return 0;
  const int Ho = (H + h->stride - 1) / h->stride, Wo = (W + h->stride - 1) / h->stride;
  const long long tiles = (long long)n * ((Wo + v->tw - 1) / v->tw) * ((Ho + 7) / 8);
  const int nsg = irb_groups(h, tiles);
  return nsg > 1 ? (size_t)nsg * n * Ho * Wo * h->cout * sizeof(float) : 0;
}

// 1 when v3d_irb_nhwc_f32 has a kernel for this block at input size H x W
extern "C" int v3d_irb_supported(const v3d_irb_weights* h, int H, int W) {
  return h && H >= 1 && W >= 1 && irb_pick(h, H, W) != nullptr;
}

extern "C" int v3d_irb_nhwc_f32(const v3d_irb_weights* h, const float* x, int n, int H, int W, float* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(h && !h->stem && x && out, V3D_ERR_BAD_ARG, "v3d_irb_nhwc_f32: null argument (or a stem handle: v3d_stem_block_f32)");
  return v3d_irb_launch(h, x, n, H, W, 0, 0, out, workspace, workspace_bytes, stream);
}

int v3d_irb_launch(const v3d_irb_weights* h, const float* x, int n, int H, int W, int ih, int iw, float* out, void* workspace,
                   size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(h && x && out, V3D_ERR_BAD_ARG, "v3d_irb_nhwc_f32: null argument");
  V3D_REQUIRE(n >= 0 && H >= 1 && W >= 1, V3D_ERR_BAD_SHAPE, "v3d_irb_nhwc_f32: n=%d H=%d W=%d", n, H, W);
  V3D_REQUIRE((reinterpret_cast<size_t>(x) & 15) == 0, V3D_ERR_BAD_ARG, "v3d_irb_nhwc_f32: x must be 16-byte aligned");
  if (n == 0) return V3D_OK;
  const IrbVariant* v = irb_pick(h, H, W);
  V3D_REQUIRE(v, V3D_ERR_UNSUPPORTED, "v3d_irb_nhwc_f32: no kernel for k=%d stride=%d cin=%d cout=%d at %d x %d (v3d_irb_supported)",
              h->ks, h->stride, h->cin, h->cout, H, W);
  IrbParams p;
  p.x = x; p.out = out;
  p.xw = h->dev + h->xw_ofs; p.dw = h->dev + h->dw_ofs; p.pw = h->dev + h->pw_ofs; p.bp = reinterpret_cast<const float*>(h->dev + h->bp_ofs);
  p.n = n; p.H = H; p.W = W; p.Ho = (H + h->stride - 1) / h->stride; p.Wo = (W + h->stride - 1) / h->stride;
  p.cin = h->cin; p.cout = h->cout; p.ns = h->ns; p.residual = h->residual; p.ih = ih; p.iw = iw;
  p.tiles_x = (p.Wo + v->tw - 1) / v->tw; p.tiles_y = (p.Ho + 7) / 8;
  const long long tiles = (long long)n * p.tiles_x * p.tiles_y;
  p.nsg = irb_groups(h, tiles);
  p.part = reinterpret_cast<float*>(workspace);
  const size_t need = p.nsg > 1 ? (size_t)p.nsg * n * p.Ho * p.Wo * h->cout * sizeof(float) : 0;
  V3D_REQUIRE(workspace_bytes >= need && (need == 0 || (workspace && (reinterpret_cast<size_t>(workspace) & 15) == 0)), V3D_ERR_BAD_ARG,
              "v3d_irb_nhwc_f32: workspace of %zu bytes, %zu needed (v3d_irb_workspace_bytes), 16-byte aligned", workspace_bytes, need);
  const long long blocks = tiles * p.nsg;
  V3D_REQUIRE(blocks < (1ll << 31) && (long long)n * H * W * (h->cin > h->cout ? h->cin : h->cout) < (1ll << 40), V3D_ERR_BAD_SHAPE,
              "v3d_irb_nhwc_f32: %lld tiles", blocks);
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set[64][sizeof(kIrbVariants) / sizeof(kIrbVariants[0])] = {};
  int dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "device ordinal %d", dev);
  const int vi = (int)(v - kIrbVariants);
  if (!attr_set[dev][vi]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)v->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, v->lds));
    attr_set[dev][vi] = true;
  }
  v3d::TimedScope ts(h->stem ? "backbone_stem_block" : "backbone_block", s);
  v->kernel<<<(unsigned)blocks, 256, v->lds, s>>>(p);
  V3D_CHECK_LAUNCH("irb_kernel");
  if (p.nsg > 1) {
    const size_t total = (size_t)n * p.Ho * p.Wo * h->cout;
    irb_reduce_kernel<<<(unsigned)((total / 4 + 255) / 256), 256, 0, s>>>(p.part, p.nsg, total, p.bp, h->residual ? x : nullptr, h->cout, out);
    V3D_CHECK_LAUNCH("irb_reduce_kernel");
  }
  return V3D_OK;
}
