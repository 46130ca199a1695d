// the 2D feature extractor of MVSNet (mv3d/subnetworks/mvsnet.py:55-105: torchvision's MnasNet-1.0 trunk
// + FeaturePyramidNetwork) as hand-written kernels.  On stock MIOpen fp32 convolutions the 71 images of a 64-view cfg2 / cfg3
// batch took 8.0 ms -- 2.4x the cost-volume step they feed -- in ~150 launches of small-channel NCHW convolutions, BatchNorm and
// ReLU passes.  Here:
//
//   * activations are CHANNELS-LAST fp32 [n, H, W, C] (C a multiple of 8): a position's channels are one contiguous run, so a
//     1x1 convolution is a plain row-major GEMM  out[P, Cout] = X[P, Cin] W[Cin, Cout]  and a depthwise tap is one float4;
//   * eval-mode BatchNorm is folded into the weights / a bias on the host; bias, ReLU and the residual (the inverted-residual
//     skip, or the FPN's nearest-upsampled top-down map) live in the epilogue of the kernel that produces the tensor;
//   * conv_gemm_kernel: 1x1 and 3x3 (FPN output) convolutions on v_mfma_f32_32x32x2_f32 -- EXACT fp32 products (the layers are
//     memory-bound: 20-150 FLOP per byte moved, so the 157 TFLOP/s fp32 matrix rate is not the limiter and no operand
//     splitting is needed).  MFMA rows are 32 consecutive positions, columns 32 output channels: a lane's accumulator
//     registers hold 16 positions of ONE channel, so the 32 lanes of a half-wave store 128 contiguous bytes per position;
//   * depthwise_kernel: k x k (3 | 5), stride 1 | 2, one thread per (output position, 4 channels);
//   * stem_kernel: the 3 -> 32 stride-2 convolution straight from the NCHW image.
// 55 launches per forward, no intermediate in another layout, one HBM round trip per tensor.
//
// Round 6: these per-layer kernels are the EXACT-FP32 path of the backbone (NativeBackbone(precision='fp32')) and the path of
// block shapes without a fused instance; the default path runs an inverted-residual block (csrc/irb.hip), the stem's three
// layers (same kernel) and a fine pyramid level (csrc/fpn.hip) as ONE kernel each on split-bf16 matrix operands: 3.33 -> 1.44 ms.
#include <vector>

#include "v3d_common.h"

struct v3d_conv_weights {
  int cout, k, ncb, nsteps;      // output channels, K = taps * Cin, column blocks of 32, K steps of 8
  float* dev;                    // [ncb][nsteps][64 lanes][4] fragments, then bias [ncb * 32]
  size_t bias_ofs;
};

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
  const float* x;        // [n, H, W, Cin]
  const float* wp;       // packed fragments
  const float* bias;     // [ncb * 32]
  const float* res;      // residual: [n, H, W, Cout] (mode 1) or [n, ceil(H / 2), ceil(W / 2), Cout] (mode 2) or null
  float* out;            // [n, H, W, Cout]
  int n, H, W, cin, cout, ncb, nsteps, relu, res_mode;
  unsigned m_hw, m_w;    // v3d::magic_u32 of H * W and W (the upsampled residual's address: two divisions per element otherwise)
  long long P;           // n * H * W
};

// out[p, co] = act(sum_{tap, c} x[p + tap, c] w[tap * Cin + c, co] + bias[co]) (+ residual)
// Workgroup: 4 waves x 32 positions; every wave NB column blocks of 32 output channels (blockIdx.y picks the group of NB).
// K runs in steps of 8 channels = 4 matrix instructions: lane (kk = lane >> 5, i = lane & 31) holds channels 8 s + 4 kk .. + 3 of
// position i (one float4 load) and the weights of the same four k for its column (one float4 per column block, packed on
// the host in exactly this order).
// KSPLIT = true (the low-resolution layers: few positions, K up to 1152): the four waves of a workgroup share ONE row block of 32
// positions and take a quarter of the K steps each -- four times the waves for the same tensor and a dependent chain a quarter
// as long (a 1152 -> 192 layer at 8 x 10 is 144 steps of four 64-cycle matrix instructions per wave otherwise, on a chip that the
// 5 680 positions of 71 images fill once with one wave per SIMD) -- the partial sums meet in LDS, wave 0 finishes.
template <int NB, int TAPS, bool KSPLIT>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvParams p) {
  __shared__ float red[KSPLIT ? 3 * NB * 16 * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 5, i = lane & 31;
  const long long p0 = KSPLIT ? (long long)blockIdx.x * 32 : ((long long)blockIdx.x * 4 + wave) * 32;
  if (p0 >= p.P) return;
  const int cb0 = blockIdx.y * NB;
  const long long pos = min(p0 + i, p.P - 1);              // (rows beyond the tensor repeat the last position; never stored)
  int py = 0, px = 0;
  if (TAPS == 9) {       // (host: P < 2^31)
    const unsigned rem = (unsigned)pos % (unsigned)(p.H * p.W);
    py = (int)(rem / (unsigned)p.W); px = (int)(rem % (unsigned)p.W);
  }
  const float* const xrow = p.x + pos * p.cin + 4 * kk;
  // (a group of NB column blocks may reach past the last one: those columns re-read the last block's weights and are never stored)
  const f32x4* wlb[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) wlb[b] = reinterpret_cast<const f32x4*>(p.wp) + (size_t)min(cb0 + b, p.ncb - 1) * p.nsteps * 64 + lane;
  const int csteps = p.cin >> 3;                             // K steps per tap
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  auto load_a = [&](int s) __attribute__((always_inline)) {
    if (TAPS == 1) return *reinterpret_cast<const f32x4*>(xrow + 8 * s);
    const int tap = s / csteps, cs = s - tap * csteps;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const bool ok = (unsigned)(py + dy) < (unsigned)p.H && (unsigned)(px + dx) < (unsigned)p.W;
    const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + (ok ? ((long long)dy * p.W + dx) * p.cin : 0) + 8 * cs);
    return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // this wave's K steps [s0, s1); operands PF steps ahead of the matrix instructions that consume them (two in the split-K
  // kernel, whose waves run one per SIMD; one in the bandwidth-bound layers, where the registers buy occupancy instead)
  constexpr int PF = KSPLIT ? 2 : 1;
  const int s0 = KSPLIT ? (p.nsteps * wave) / 4 : 0, s1 = KSPLIT ? (p.nsteps * (wave + 1)) / 4 : p.nsteps;
  f32x4 a_q[PF], w_q[PF][NB];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    const int sd = min(s0 + d, s1 - 1);
    a_q[d] = load_a(sd);
#pragma unroll
    for (int b = 0; b < NB; ++b) w_q[d][b] = wlb[b][(size_t)sd * 64];
  }
  for (int s = s0; s < s1; s += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      if (s + d >= s1) break;
      const f32x4 a = a_q[d];
      f32x4 w[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) w[b] = w_q[d][b];
      if (s + d + PF < s1) {                                    // (wave-uniform)
        const int sn = s + d + PF;
        a_q[d] = load_a(sn);
#pragma unroll
        for (int b = 0; b < NB; ++b) w_q[d][b] = wlb[b][(size_t)sn * 64];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], w[b][m], acc[b], 0, 0, 0);
    }
  }
  if (KSPLIT) {
    if (wave > 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wave - 1) * NB + b) * 16 + r) * 64 + lane] = acc[b][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int wv = 0; wv < 3; ++wv)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] += red[((wv * NB + b) * 16 + r) * 64 + lane];
  }
  // epilogue: lane (g = kk, column i) holds rows 8 j + 4 g + r of its column
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int co = (cb0 + b) * 32 + i;
    if (co >= p.cout) continue;
    const float bs = p.bias[co];
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) {
      const long long q = p0 + 8 * (r16 >> 2) + 4 * kk + (r16 & 3);
      if (q >= p.P) continue;
      float v = acc[b][r16] + bs;
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.res_mode == 1) v += p.res[q * p.cout + co];
      else if (p.res_mode == 2) {
        const unsigned hw = (unsigned)(p.H * p.W), img = v3d::udiv_magic((unsigned)q, hw, p.m_hw), rem = (unsigned)q - img * hw;
        const unsigned y = v3d::udiv_magic(rem, (unsigned)p.W, p.m_w), x = rem - y * (unsigned)p.W;
        // (the coarser map has ceil(H / 2) x ceil(W / 2) positions -- a stride-2 / pad k/2 convolution's output -- and nearest
        // interpolation to H x W reads floor(y * ceil(H / 2) / H) = y >> 1 for even AND odd H: mvsnet.py:86-88, FPN top-down)
        v += p.res[((size_t)(img * (unsigned)((p.H + 1) >> 1) + (y >> 1)) * (unsigned)((p.W + 1) >> 1) + (x >> 1)) * p.cout + co];
      }
      p.out[q * p.cout + co] = v;
    }
  }
}

// depthwise k x k convolution, padding k / 2, stride S, + bias (+ ReLU): one thread per (NX consecutive output columns, 4
// channels): the NX outputs share their input columns (k = 5, stride 1: 8 x 5 loads for 4 outputs instead of 100), and the bounds
// tests / address arithmetic are per input column instead of per tap
template <int KS, int S, int NX>
__global__ __launch_bounds__(256) void depthwise_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int n, int H,
                                                        int W, int C, int relu) {
  constexpr int NIN = (NX - 1) * S + KS;           // input columns the NX outputs touch
  const int c4n = C >> 2;
  const int Ho = (H + S - 1) / S, Wo = (W + S - 1) / S, Wg = (Wo + NX - 1) / NX;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)n * Ho * Wg * c4n) return;
  const int c4 = (int)(t % c4n);
  const long long q = t / c4n;
  const int xg = (int)(q % Wg), yo = (int)((q / Wg) % Ho), img = (int)(q / ((long long)Wg * Ho));
  const int xo0 = xg * NX, xi0 = xo0 * S - KS / 2;
  const f32x4* const xi = reinterpret_cast<const f32x4*>(x + (size_t)img * H * W * C) + c4;
  const f32x4* const w4 = reinterpret_cast<const f32x4*>(w) + c4;            // [KS * KS][C]
  const f32x4 b4 = reinterpret_cast<const f32x4*>(bias)[c4];
  f32x4 acc[NX];
#pragma unroll
  for (int o = 0; o < NX; ++o) acc[o] = b4;
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
    const int yy = yo * S + ky - KS / 2;
    if ((unsigned)yy >= (unsigned)H) continue;
    f32x4 wr[KS];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) wr[kx] = w4[(ky * KS + kx) * c4n];
    const f32x4* const row = xi + (size_t)yy * W * c4n;
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      const int xx = xi0 + j;
      const f32x4 v = (unsigned)xx < (unsigned)W ? row[(size_t)xx * c4n] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < NX; ++o) {
        const int kx = j - o * S;
        if (kx >= 0 && kx < KS) acc[o] = __builtin_elementwise_fma(v, wr[kx], acc[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < NX; ++o) {
    if (xo0 + o >= Wo) break;
    f32x4 v = acc[o];
    if (relu) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
    reinterpret_cast<f32x4*>(out)[(((size_t)img * Ho + yo) * Wo + xo0 + o) * c4n + c4] = v;
  }
}

// stem: Conv2d(3 -> 32, k3, stride 2, pad 1) + folded BatchNorm + ReLU from the NCHW image to channels-last [n, H/2, W/2, 32];
// one thread per output position, all 32 channels: the 27 x 32 weights are wave-uniform (scalar loads, SGPR operands of the FMAs)
// and a thread stores its position's 128 bytes.  (One thread per (position, 8 channels) read its 216 weights with vector loads:
// 0.20 ms for 256 MB.)  The taps come from clamped addresses and are masked (`ok ? img[...] : 0` puts every load behind its own
// branch and wait).
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int n, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= (long long)n * Ho * Wo) return;
  const int xo = (int)(q % Wo), yo = (int)((q / Wo) % Ho), im = (int)(q / ((long long)Wo * Ho));
  float v[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = 2 * yo + ky - 1, yc = min(max(yy, 0), H - 1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = 2 * xo + kx - 1, xc = min(max(xx, 0), W - 1);
        const unsigned keep = 0u - (unsigned)((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W);
        v[(c * 3 + ky) * 3 + kx] = __uint_as_float(__float_as_uint(img[(((size_t)im * 3 + c) * H + yc) * W + xc]) & keep);
      }
    }
  f32x4* const o = reinterpret_cast<f32x4*>(out + q * 32);
#pragma unroll
  for (int c8 = 0; c8 < 4; ++c8) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = bias[c8 * 8 + k];
#pragma unroll
    for (int t9 = 0; t9 < 27; ++t9) {
      const float* wt = w + t9 * 32 + c8 * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(v[t9], wt[k], acc[k]);
    }
    o[2 * c8] = (f32x4){fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
    o[2 * c8 + 1] = (f32x4){fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f)};
  }
}

// [n, HW, C] -> [n, C, HW] (the reference layout of the feature maps the rest of the path consumes); C a multiple of 32
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  __shared__ float tile[32][33];
  const int img = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int pp = p0 + r;
    tile[r][tx] = pp < HW ? in[((size_t)img * HW + pp) * C + c0 + tx] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int pp = p0 + tx;
    if (pp < HW) out[((size_t)img * C + c0 + r) * HW + pp] = tile[tx][r];
  }
}

}  // namespace

// weight [Cout, K] row-major on the HOST (K = taps * Cin, tap-major: k = tap * Cin + c, i.e. Conv2d weight [Cout, Cin, kh, kw]
// permuted to [Cout, kh, kw, Cin]), BatchNorm already folded; bias [Cout] or null
extern "C" int v3d_conv_pack(const float* w_host, const float* bias_host, int cout, int k, v3d_conv_weights** out_handle) {
  V3D_REQUIRE(w_host && out_handle && cout >= 1 && k >= 8 && k % 8 == 0, V3D_ERR_BAD_SHAPE,
              "v3d_conv_pack: cout=%d, K=%d (K must be a multiple of 8)", cout, k);
  v3d_conv_weights* h = new v3d_conv_weights();
  h->cout = cout; h->k = k; h->ncb = (cout + 31) / 32; h->nsteps = k / 8;
  h->bias_ofs = (size_t)h->ncb * h->nsteps * 256;
  std::vector<float> host(h->bias_ofs + (size_t)h->ncb * 32, 0.f);
  for (int cb = 0; cb < h->ncb; ++cb)
    for (int s = 0; s < h->nsteps; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int m = 0; m < 4; ++m) {
          const int co = cb * 32 + (lane & 31), kq = 8 * s + 4 * (lane >> 5) + m;
          host[(((size_t)cb * h->nsteps + s) * 64 + lane) * 4 + m] = co < cout ? w_host[(size_t)co * k + kq] : 0.f;
        }
  for (int co = 0; co < cout; ++co) host[h->bias_ofs + co] = bias_host ? bias_host[co] : 0.f;
  hipError_t e = hipMalloc((void**)&h->dev, host.size() * sizeof(float));
  if (e != hipSuccess) { delete h; return v3d::fail(V3D_ERR_HIP, "hipMalloc(conv weights): %s", hipGetErrorString(e)); }
  e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(h->dev); delete h; return v3d::fail(V3D_ERR_HIP, "hipMemcpy(conv weights): %s", hipGetErrorString(e)); }
  *out_handle = h;
  return V3D_OK;
}

extern "C" void v3d_conv_free(v3d_conv_weights* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  delete h;
}

extern "C" int v3d_conv_nhwc_f32(const v3d_conv_weights* h, const float* x, int n, int H, int W, int cin, int taps, int relu,
                                 int res_mode, const float* res, float* out, void* stream) {
  V3D_REQUIRE(h && x && out, V3D_ERR_BAD_ARG, "v3d_conv_nhwc_f32: null argument");
  V3D_REQUIRE((taps == 1 || taps == 9) && cin % 8 == 0 && taps * cin == h->k, V3D_ERR_BAD_SHAPE,
              "v3d_conv_nhwc_f32: taps=%d cin=%d against packed K=%d", taps, cin, h->k);
  V3D_REQUIRE(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || res), V3D_ERR_BAD_ARG, "v3d_conv_nhwc_f32: residual mode %d", res_mode);
  V3D_REQUIRE((reinterpret_cast<size_t>(x) & 15) == 0, V3D_ERR_BAD_ARG, "v3d_conv_nhwc_f32: x must be 16-byte aligned");
  const long long P = (long long)n * H * W;
  if (P == 0) return V3D_OK;
  V3D_REQUIRE(P < (1ll << 31) && P * (long long)(cin > h->cout ? cin : h->cout) < (1ll << 40), V3D_ERR_BAD_SHAPE,
              "v3d_conv_nhwc_f32: %lld positions", P);
  ConvParams p;
  p.x = x; p.wp = h->dev; p.bias = h->dev + h->bias_ofs; p.res = res; p.out = out;
  p.n = n; p.H = H; p.W = W; p.cin = cin; p.cout = h->cout; p.ncb = h->ncb; p.nsteps = h->nsteps; p.relu = relu; p.res_mode = res_mode; p.P = P;
  p.m_hw = v3d::magic_u32((unsigned long long)P, (unsigned)(H * W)); p.m_w = v3d::magic_u32((unsigned long long)H * W, (unsigned)W);
  hipStream_t s = (hipStream_t)stream;
  // Few positions and a long K (the 1/16 and 1/32 resolution layers): split K over the four waves of a workgroup.  Otherwise a
  // workgroup holds 128 positions and every wave as many column blocks as keep >= ~2 workgroups per CU in flight (a wide tile
  // re-reads the input less often).
  const long long rows = (P + 127) / 128;
  v3d::TimedScope ts(taps == 9 ? "backbone_conv3x3" : "backbone_conv1x1", s);
  if (rows * h->ncb < 1024 && h->nsteps >= 32) {
    const int nb = h->ncb >= 2 && ((P + 31) / 32) * ((h->ncb + 1) / 2) >= 1024 ? 2 : 1;
    const dim3 grid((unsigned)((P + 31) / 32), (unsigned)((h->ncb + nb - 1) / nb));
    if (taps == 1) { if (nb == 2) conv_gemm_kernel<2, 1, true><<<grid, 256, 0, s>>>(p); else conv_gemm_kernel<1, 1, true><<<grid, 256, 0, s>>>(p); }
    else { if (nb == 2) conv_gemm_kernel<2, 9, true><<<grid, 256, 0, s>>>(p); else conv_gemm_kernel<1, 9, true><<<grid, 256, 0, s>>>(p); }
  } else {
    int nb = 1;
    for (int c = h->ncb < 4 ? h->ncb : 4; c >= 2; --c)
      if (rows * ((h->ncb + c - 1) / c) >= 512) { nb = c; break; }
    const dim3 grid((unsigned)rows, (unsigned)((h->ncb + nb - 1) / nb));
#define V3D_CG(NB_)                                                             \
  do {                                                                          \
    if (taps == 1) conv_gemm_kernel<NB_, 1, false><<<grid, 256, 0, s>>>(p);     \
    else conv_gemm_kernel<NB_, 9, false><<<grid, 256, 0, s>>>(p);               \
  } while (0)
    if (nb == 4) V3D_CG(4); else if (nb == 3) V3D_CG(3); else if (nb == 2) V3D_CG(2); else V3D_CG(1);
#undef V3D_CG
  }
  V3D_CHECK_LAUNCH("conv_gemm_kernel");
  return V3D_OK;
}

// w [k*k][C] (tap-major, BatchNorm folded) and bias [C] on the DEVICE
extern "C" int v3d_depthwise_nhwc_f32(const float* x, const float* w, const float* bias, int n, int H, int W, int C, int ksize,
                                      int stride, int relu, float* out, void* stream) {
  V3D_REQUIRE(x && w && bias && out, V3D_ERR_BAD_ARG, "v3d_depthwise_nhwc_f32: null argument");
  V3D_REQUIRE((ksize == 3 || ksize == 5) && (stride == 1 || stride == 2) && C % 4 == 0, V3D_ERR_UNSUPPORTED,
              "v3d_depthwise_nhwc_f32: k=%d stride=%d C=%d", ksize, stride, C);
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const int nx = stride == 1 ? 4 : 2;
  const long long threads = (long long)n * Ho * ((Wo + nx - 1) / nx) * (C / 4);
  if (threads == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  v3d::TimedScope ts("backbone_depthwise", s);
  const unsigned grid = (unsigned)((threads + 255) / 256);
  if (ksize == 3 && stride == 1) depthwise_kernel<3, 1, 4><<<grid, 256, 0, s>>>(x, w, bias, out, n, H, W, C, relu);
  else if (ksize == 3) depthwise_kernel<3, 2, 2><<<grid, 256, 0, s>>>(x, w, bias, out, n, H, W, C, relu);
  else if (stride == 1) depthwise_kernel<5, 1, 4><<<grid, 256, 0, s>>>(x, w, bias, out, n, H, W, C, relu);
  else depthwise_kernel<5, 2, 2><<<grid, 256, 0, s>>>(x, w, bias, out, n, H, W, C, relu);
  V3D_CHECK_LAUNCH("depthwise_kernel");
  return V3D_OK;
}

// image [n, 3, H, W] (NCHW, H and W even), w [27][32] tap-major (c, ky, kx) and bias [32] on the device -> [n, H/2, W/2, 32]
extern "C" int v3d_stem_f32(const float* image, const float* w, const float* bias, int n, int H, int W, float* out, void* stream) {
  V3D_REQUIRE(image && w && bias && out, V3D_ERR_BAD_ARG, "v3d_stem_f32: null argument");
  V3D_REQUIRE(H % 2 == 0 && W % 2 == 0, V3D_ERR_BAD_SHAPE, "v3d_stem_f32: H, W must be even");
  const long long threads = (long long)n * (H / 2) * (W / 2);
  if (threads == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  v3d::TimedScope ts("backbone_stem", s);
  stem_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(image, w, bias, out, n, H, W);
  V3D_CHECK_LAUNCH("stem_kernel");
  return V3D_OK;
}

extern "C" int v3d_nhwc_to_nchw_f32(const float* in, float* out, int n, int C, int HW, void* stream) {
  V3D_REQUIRE(in && out, V3D_ERR_BAD_ARG, "v3d_nhwc_to_nchw_f32: null argument");
  V3D_REQUIRE(C % 32 == 0 && n >= 0 && n < 65536, V3D_ERR_UNSUPPORTED, "v3d_nhwc_to_nchw_f32: C=%d (multiple of 32), n=%d", C, n);
  if (n == 0 || HW == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  v3d::TimedScope ts("backbone_to_nchw", s);
  nhwc_to_nchw_kernel<<<dim3((unsigned)((HW + 31) / 32), (unsigned)(C / 32), (unsigned)n), 256, 0, s>>>(in, out, C, HW);
  V3D_CHECK_LAUNCH("nhwc_to_nchw_kernel");
  return V3D_OK;
}
