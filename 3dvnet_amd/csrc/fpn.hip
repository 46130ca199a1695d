// one level of the feature pyramid (mv3d/subnetworks/mvsnet.py:83-105; torchvision
// FeaturePyramidNetwork: inner = lateral 1x1 (C_l) + nearest-upsampled inner of the coarser level; out = 3x3 (inner), both with bias)
// as ONE kernel for the three fine levels, which carry the pyramid's cost: at 128 x 160 the round-5 path took 0.12 ms for the
// lateral convolution, 0.34 ms for the 3x3 on exact-fp32 matrix instructions (27 GFLOP at 80 TFLOP/s: bound by that pipe) and 0.03 ms
// to turn the channels-last result into the reference layout.  Here a workgroup walks tiles of 8 x 16 positions:
//
//   X     inner[region position, 32] = x[position, 0:cin] Wl + bl + inner_coarser[position >> 1]     (matrix cores, split-bf16)
//         for the tile's 10 x 18 region, written to LDS as split-bf16 planes (zero outside the image: the 3x3's padding) and,
//         for the tile's own positions, to HBM as the next finer level's top-down input;
//   conv  out[tile position, 32] = sum over 9 taps x 32 channels of inner[position + tap] W3 + b3   (matrix cores, split-bf16)
//         -- a tap is a row offset into the planes, every operand one 16-byte LDS read -- stored straight in the reference
//         layout [n, 32, H, W] (matrix rows = channels, columns = positions: a register's 32 lanes are two 64-byte runs).
//
// The weights (36 KB of 3x3 fragments + the lateral fragments) are loaded into LDS once per workgroup; the next tile's input rows
// and top-down values are requested before the current tile's convolution.  feat_dim = 32, cin <= 48 (the levels at 1/2, 1/4, 1/8);
// the two coarse levels stay on the per-layer kernels (0.05 ms together).
#include <cstring>
#include <vector>

#include "v3d_common.h"

struct v3d_fpn_weights {
  int cin, csteps;
  char* dev;            // lateral fragments [csteps][hi, lo][64][16 B] | 3x3 fragments [9 taps][2 steps][hi, lo][64][16 B] | bl [32] | b3 [32]
  size_t w3_ofs, bias_ofs;
};

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTH = 8, kTW = 16, kRH = kTH + 2, kRW = kTW + 2, kR = kRH * kRW;     // tile, region (180 positions)
constexpr int kRows = 192;                                   // region rows + the padding rows of the sixth row block
constexpr int kPlane = kRows * 16;                           // bytes: [row][8 bf16]; plane = 16 st + 8 g' channel chunk
constexpr int kEH = 0, kEL = 4 * kPlane, kW3 = 8 * kPlane, kW3Bytes = 18 * 2048;
constexpr int kWL = kW3 + kW3Bytes;

struct FpnParams {
  const float* x;          // [n, H, W, cin]
  const float* coarse;     // [n, ceil(H/2), ceil(W/2), 32] or null
  float* inner;            // [n, H, W, 32] or null
  float* out;              // [n, 32, H, W]
  const char* w;           // packed image
  int n, H, W, cin, tiles_x, tiles_y, n_tiles;
  size_t w3_ofs, bias_ofs;
};

__device__ __forceinline__ unsigned fpn_pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2_));
}
__device__ __forceinline__ void fpn_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = fpn_pack_bf16x2(a, b);
  lo = fpn_pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

template <int CSTEPS>
__global__ __launch_bounds__(256, 2) void fpn_level_kernel(FpnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const EH = smem + kEH;
  unsigned char* const EL = smem + kEL;
  const unsigned char* const W3 = smem + kW3;
  const unsigned char* const WL = smem + kWL;
  // (the biases are read from global memory where they are used: a 16-byte LDS read must not feed vector instructions beside
  // matrix instructions in flight)
  const float* const BIAS = reinterpret_cast<const float*>(p.w + p.bias_ofs);                // bl [32], b3 [32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, n = lane & 31;
  const int Hc = (p.H + 1) >> 1, Wc = (p.W + 1) >> 1;

  // weights -> LDS, once
  for (int k = tid; k < kW3Bytes / 16; k += 256) reinterpret_cast<u32x4*>(smem + kW3)[k] = reinterpret_cast<const u32x4*>(p.w + p.w3_ofs)[k];
  for (int k = tid; k < CSTEPS * 128; k += 256) reinterpret_cast<u32x4*>(smem + kWL)[k] = reinterpret_cast<const u32x4*>(p.w)[k];

  // X role: this wave's region row blocks wave and wave + 4 (< 6); lane (g, n) = region position 32 rb + n
  constexpr int NRB = 2;
  struct Pos { int y, x; bool ok, own; };
  auto region_pos = [&](int tile, int q) __attribute__((always_inline)) {
    const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x, ty = t2 % p.tiles_y;
    const int r = (wave + 4 * q) * 32 + n, ry = r / kRW, rx = r - ry * kRW;
    Pos o;
    o.y = ty * kTH - 1 + ry;
    o.x = tx * kTW - 1 + rx;
    o.ok = r < kR && (unsigned)o.y < (unsigned)p.H && (unsigned)o.x < (unsigned)p.W;
    o.own = o.ok && ry >= 1 && ry <= kTH && rx >= 1 && rx <= kTW;
    return o;
  };
  f32x4 araw[NRB][CSTEPS][2], td[NRB][4];
  auto prefetch = [&](int tile) __attribute__((always_inline)) {
    const int img = tile / (p.tiles_x * p.tiles_y);
#pragma unroll
    for (int q = 0; q < NRB; ++q) {
      if (wave + 4 * q >= 6) continue;                       // (wave-uniform)
      const Pos o = region_pos(tile, q);
      const int yc = min(max(o.y, 0), p.H - 1), xc = min(max(o.x, 0), p.W - 1);
      const float* const xr = p.x + ((size_t)(img * p.H + yc) * p.W + xc) * p.cin;
#pragma unroll
      for (int st = 0; st < CSTEPS; ++st) {
        const int k0 = 16 * st + 8 * g, ko = k0 < p.cin ? k0 : 0;
        araw[q][st][0] = *reinterpret_cast<const f32x4*>(xr + ko);
        araw[q][st][1] = *reinterpret_cast<const f32x4*>(xr + ko + 4);
      }
      if (p.coarse) {
        const float* const cr = p.coarse + ((size_t)(img * Hc + (yc >> 1)) * Wc + (xc >> 1)) * 32 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) td[q][j] = *reinterpret_cast<const f32x4*>(cr + 8 * j);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) td[q][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // (an XCD's workgroups sit on consecutive tiles: shared halos and top-down rows meet in its L2)
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n_tiles);
  int tile = walk.t;
  if (tile < walk.end) prefetch(tile);
  __syncthreads();
#pragma unroll 1
  for (; tile < walk.end; tile += walk.step) {
    const int img = tile / (p.tiles_x * p.tiles_y);
    // ---- X: lateral 1x1 + bias + top-down -> split planes in LDS (+ the level's inner map) ----------------------------------------
#pragma unroll
    for (int q = 0; q < NRB; ++q) {
      const int rb = wave + 4 * q;
      if (rb >= 6) continue;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int st = 0; st < CSTEPS; ++st) {
        const f32x4 a0 = araw[q][st][0], a1 = araw[q][st][1];
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        fpn_split2(a0.x, a0.y, h0, l0);
        fpn_split2(a0.z, a0.w, h1, l1);
        fpn_split2(a1.x, a1.y, h2, l2);
        fpn_split2(a1.z, a1.w, h3, l3);
        const unsigned keep = 16 * st + 8 * g < p.cin ? 0xffffffffu : 0u;
        const bf16x8 xh = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3} & (u32x4){keep, keep, keep, keep});
        const bf16x8 xl = __builtin_bit_cast(bf16x8, (u32x4){l0, l1, l2, l3} & (u32x4){keep, keep, keep, keep});
        const bf16x8 wh = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(WL + (st * 2 + 0) * 1024)[lane]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(WL + (st * 2 + 1) * 1024)[lane]);
        // rows = inner channels, columns = positions
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc, 0, 0, 0);
      }
      // lane (g, n) holds channels 8 j + 4 g .. + 3 of region position 32 rb + n
      const Pos o = region_pos(tile, q);
      const int r = rb * 32 + n;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bl = *reinterpret_cast<const f32x4*>(BIAS + 8 * j + 4 * g);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o.ok ? acc[4 * j + e] + bl[e] + td[q][j][e] : 0.f;
        if (p.inner && o.own) *reinterpret_cast<f32x4*>(p.inner + ((size_t)(img * p.H + o.y) * p.W + o.x) * 32 + 8 * j + 4 * g) = v;
        unsigned h01, l01, h23, l23;
        fpn_split2(v.x, v.y, h01, l01);
        fpn_split2(v.z, v.w, h23, l23);
        *reinterpret_cast<u32x2*>(EH + (j * kRows + r) * 16 + g * 8) = (u32x2){h01, h23};
        *reinterpret_cast<u32x2*>(EL + (j * kRows + r) * 16 + g * 8) = (u32x2){l01, l23};
      }
    }
    __syncthreads();
    // the next tile's rows on their way during the convolution
    const int next = tile + walk.step;
    if (next < walk.end) prefetch(next);
    // ---- conv: 3x3 over the planes; wave = 32 tile positions (two tile rows), rows = output channels ------------------------------
    {
      const int t = wave * 32 + n, ty = t >> 4, tx = t & 15;
      const int base = ty * kRW + tx;
      f32x16 acc[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int row = base + (tap / 3) * kRW + tap % 3;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(EH + ((st * 2 + g) * kRows + row) * 16));
          const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(EL + ((st * 2 + g) * kRows + row) * 16));
          const bf16x8 wh = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(W3 + ((tap * 2 + st) * 2 + 0) * 1024)[lane]);
          const bf16x8 wl = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(W3 + ((tap * 2 + st) * 2 + 1) * 1024)[lane]);
          f32x16& a = acc[st];
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, a, 0, 0, 0);
        }
      }
      const int tx0 = (tile % p.tiles_x) * kTW, ty0 = ((tile / p.tiles_x) % p.tiles_y) * kTH;
      const int y = ty0 + ty, x = tx0 + tx;
      if (y < p.H && x < p.W) {
        float* const o = p.out + ((size_t)img * 32 * p.H + y) * p.W + x;
        const size_t cs = (size_t)p.H * p.W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b3 = *reinterpret_cast<const f32x4*>(BIAS + 32 + 8 * j + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[(size_t)(8 * j + 4 * g + e) * cs] = acc[0][4 * j + e] + acc[1][4 * j + e] + b3[e];
        }
      }
    }
    __syncthreads();
  }
}

unsigned fpn_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
float fpn_bf16_value(unsigned h) {
  const unsigned u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace

// HOST weights: w_lateral [32, cin] (+ b_lateral [32]), w_out [32, 32, 3, 3] (+ b_out [32]); cin a multiple of 8, <= 48
extern "C" int v3d_fpn_pack(const float* w_lateral, const float* b_lateral, const float* w_out, const float* b_out, int cin,
                            v3d_fpn_weights** out_handle) {
  V3D_REQUIRE(w_lateral && b_lateral && w_out && b_out && out_handle, V3D_ERR_BAD_ARG, "v3d_fpn_pack: null argument");
  V3D_REQUIRE(cin >= 8 && cin % 8 == 0 && cin <= 48, V3D_ERR_UNSUPPORTED, "v3d_fpn_pack: cin=%d (a multiple of 8, at most 48)", cin);
  v3d_fpn_weights* h = new v3d_fpn_weights();
  h->cin = cin; h->csteps = (cin + 15) / 16;
  h->w3_ofs = (size_t)h->csteps * 2048;
  h->bias_ofs = h->w3_ofs + kW3Bytes;
  std::vector<unsigned char> host(h->bias_ofs + 64 * 4, 0);
  auto put = [&](size_t hi_ofs, float w) {
    const unsigned hi = fpn_rne(w), lo = fpn_rne(w - fpn_bf16_value(hi));
    const unsigned short h16 = (unsigned short)hi, l16 = (unsigned short)lo;
    memcpy(&host[hi_ofs], &h16, 2);
    memcpy(&host[hi_ofs + 1024], &l16, 2);
  };
  for (int st = 0; st < h->csteps; ++st)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int m = lane & 31, k = 16 * st + 8 * (lane >> 5) + e;
        put((size_t)st * 2048 + lane * 16 + e * 2, k < cin ? w_lateral[(size_t)m * cin + k] : 0.f);
      }
  for (int tap = 0; tap < 9; ++tap)
    for (int st = 0; st < 2; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = lane & 31, ci = 16 * st + 8 * (lane >> 5) + e;
          put(h->w3_ofs + (size_t)(tap * 2 + st) * 2048 + lane * 16 + e * 2, w_out[((size_t)co * 32 + ci) * 9 + tap]);
        }
  float* const bias = reinterpret_cast<float*>(&host[h->bias_ofs]);
  for (int c = 0; c < 32; ++c) { bias[c] = b_lateral[c]; bias[32 + c] = b_out[c]; }
  hipError_t e = hipMalloc((void**)&h->dev, host.size());
  if (e != hipSuccess) { delete h; return v3d::fail(V3D_ERR_HIP, "hipMalloc(pyramid weights): %s", hipGetErrorString(e)); }
  e = hipMemcpy(h->dev, host.data(), host.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return v3d::fail(V3D_ERR_HIP, "hipMemcpy(pyramid weights): %s", hipGetErrorString(e)); }
  *out_handle = h;
  return V3D_OK;
}

extern "C" void v3d_fpn_free(v3d_fpn_weights* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  delete h;
}

// x [n, H, W, cin] channels-last; coarse_inner [n, ceil(H/2), ceil(W/2), 32] channels-last or NULL (the coarsest level);
// inner_out [n, H, W, 32] channels-last or NULL (the finest level: nobody reads it); out [n, 32, H, W] (the reference layout)
extern "C" int v3d_fpn_level_f32(const v3d_fpn_weights* h, const float* x, const float* coarse_inner, int n, int H, int W,
                                 float* inner_out, float* out, void* stream) {
  V3D_REQUIRE(h && x && out, V3D_ERR_BAD_ARG, "v3d_fpn_level_f32: null argument");
  V3D_REQUIRE(n >= 0 && H >= 1 && W >= 1, V3D_ERR_BAD_SHAPE, "v3d_fpn_level_f32: n=%d H=%d W=%d", n, H, W);
  V3D_REQUIRE((reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(coarse_inner) & 15) == 0 &&
                  (reinterpret_cast<size_t>(inner_out) & 15) == 0,
              V3D_ERR_BAD_ARG, "v3d_fpn_level_f32: tensors must be 16-byte aligned");
  if (n == 0) return V3D_OK;
  FpnParams p;
  p.x = x; p.coarse = coarse_inner; p.inner = inner_out; p.out = out; p.w = h->dev;
  p.n = n; p.H = H; p.W = W; p.cin = h->cin;
  p.tiles_x = (W + kTW - 1) / kTW; p.tiles_y = (H + kTH - 1) / kTH;
  const long long tiles = (long long)n * p.tiles_x * p.tiles_y;
  V3D_REQUIRE(tiles < (1ll << 31) && (long long)n * H * W * 48 < (1ll << 40), V3D_ERR_BAD_SHAPE, "v3d_fpn_level_f32: %lld tiles", tiles);
  p.n_tiles = (int)tiles;
  p.w3_ofs = h->w3_ofs; p.bias_ofs = h->bias_ofs;
  hipStream_t s = (hipStream_t)stream;
  const int lds = kWL + h->csteps * 2048;
  void (*kernel)(FpnParams) = h->csteps == 1 ? fpn_level_kernel<1> : h->csteps == 2 ? fpn_level_kernel<2> : fpn_level_kernel<3>;
  static bool attr_set[64][3] = {};
  int dev = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "device ordinal %d", dev);
  if (!attr_set[dev][h->csteps - 1]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set[dev][h->csteps - 1] = true;
  }
  const unsigned grid = v3d::persistent_grid(tiles, 2);
  v3d::TimedScope ts("backbone_pyramid_level", s);
  kernel<<<grid, 256, lds, s>>>(p);
  V3D_CHECK_LAUNCH("fpn_level_kernel");
  return V3D_OK;
}
