// Rows A1-A4 of SURVEY.md §8a: plane-sweep homography warp of source-view features + cross-view
// variance, fused (no world points, sampling grid, warped volume or squared volume is ever
// materialised).  Reference semantics: mv3d/utils.py:86-108, mv3d/subnetworks/mvsnet.py:192-216.
//
// Data layout in HBM
//   feat   [n_img, C, Hf, Wf]  (reference layout)  -> transposed once per call into
//   featT  [n_img, Hf+2, Wf+2, C]  (workspace, zero border) so that one bilinear tap of all C channels is one
//                               contiguous 4*C-byte run (C=32: exactly one 128-B cache line) and out-of-image taps read zeros;
//   var    [n_ref, C, D, h, w] (reference layout, consumed as-is by the 3D-conv regulariser).
//
// Work decomposition: one 256-thread workgroup per (reference view, chunk of DB depth planes,
// tile of 64 plane-grid pixels).  Per plane:
//   phase 1  every (pixel, edge) pair is projected once (256 threads, results in LDS):
//            clamped tap coordinates + the four bilinear weights with out-of-range taps zeroed
//            (== grid_sample padding_mode='zeros');
//   phase 2  C/4 lanes per pixel; each lane gathers float4 (4 channels) for the 4 taps of every
//            edge -> the C/4 lanes of a pixel read one full contiguous run per tap -- and
//            accumulates sum / sum of squares in registers in edge order (deterministic);
//   phase 3  the [C][64] tile is transposed through LDS and written as 256-B rows of `var`.
#include <cstdlib>

#include <vector>

#include "v3d_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPix = 64;     // plane-grid pixels per workgroup
constexpr int kDB = 4;       // depth planes per workgroup
constexpr int kMaxE = 8;     // edges per LDS pass

struct PsvParams {
  const float* featT;
  const float* K;
  const float* R;
  const float* t;
  const int* ref_img;
  const int* edge_ofs;
  const int* edge_src;
  float* var;
  const float* camp;   // [n_img][kCamStride] per-image camera block, see cam_setup_kernel
  int n_img, n_ref, Hf, Wf, H, W, D, h, w, n_ptile;
  double x_step, y_step, z_start, z_step, z_end;
  // window kernel: launch constants the host works out once -- per wave they were two IEEE f64 divisions and three integer
  // divisions through v_rcp_iflag_f32 (round 4: ~110 of the ~2 100 VALU issue slots of a wave at cfg2)
  float rWm1, rHm1;                        // (float)(1.0 / (double)(W - 1)), (float)(1.0 / (double)(H - 1))
  unsigned m_dchunk, m_ptile, m_w;         // v3d::magic_u32() of the plane-chunk count, of n_ptile and of w (0: divide)
};


// [n_img, C, HW] -> [n_img, HW, C]
template <int C>
__global__ __launch_bounds__(256) void transpose_channel_last_kernel(const float* __restrict__ in,
                                                                      float* __restrict__ out,
                                                                      int HW) {
  __shared__ float tile[C][kPix + 1];
  const int img = blockIdx.y;
  const int p0 = blockIdx.x * kPix;
  const float* src = in + (size_t)img * C * HW;
  float* dst = out + (size_t)img * HW * C;
  for (int i = threadIdx.x; i < C * kPix; i += 256) {
    int c = i / kPix, p = i % kPix;
    tile[c][p] = (p0 + p < HW) ? src[(size_t)c * HW + p0 + p] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * kPix; i += 256) {
    int p = i / C, c = i % C;
    if (p0 + p < HW) dst[(size_t)(p0 + p) * C + c] = tile[c][p];
  }
}

// zero cells behind the last bordered image (see transpose_bordered_kernel)
__host__ __device__ constexpr int kTailCells(int Wf) { return 2 * (Wf + 2) + 16; }

// [n_img, C, Hf, Wf] -> [n_img, Hf + 2, Wf + 2, C] with a border of zero cells (+ a tail of zero cells behind the last image).
// grid_sample's padding_mode='zeros' then needs no per-tap validity test in the warp kernels: a tap outside the image reads
// a zero cell, and a sample further out is clamped onto the border, where both of its in-range taps are zero cells and the
// other two carry weight 0 (make_taps).  One workgroup = 64 consecutive cells of one padded image.
template <int C>
__global__ __launch_bounds__(256) void transpose_bordered_kernel(const float* __restrict__ in, float* __restrict__ out, int Hf,
                                                                  int Wf, int n_img) {
  __shared__ float tile[C][kPix + 1];
  const int img = blockIdx.y;
  const int Wp = Wf + 2, HW = Hf * Wf, n_cell = (Hf + 2) * Wp;
  const int q0 = blockIdx.x * kPix;                      // 64 consecutive cells of the padded image (row-major)
  const float* src = in + (size_t)img * C * HW;
  float* dst = out + (size_t)img * n_cell * C;
  for (int i = threadIdx.x; i < C * kPix; i += 256) {
    const int c = i / kPix, q = q0 + i % kPix;
    const int yb = q / Wp, xb = q - yb * Wp;
    const bool inside = q < n_cell && yb >= 1 && yb <= Hf && xb >= 1 && xb <= Wf;
    tile[c][i % kPix] = inside ? src[(size_t)c * HW + (yb - 1) * Wf + xb - 1] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * kPix; i += 256) {
    const int px = i / C, c = i % C;
    if (q0 + px < n_cell) dst[(size_t)(q0 + px) * C + c] = tile[c][px];
  }
  // A sample clamped onto the lower border reads its weight-0 taps one padded row further down: the next image's top border,
  // or, behind the last image, this tail of zero cells.  Weight 0 times recycled memory holding a NaN pattern would be NaN.
  // The window kernel copies whole 8-cell runs of up to one row below a footprint: the tail is two bordered rows + 16 cells.
  if (img == n_img - 1 && blockIdx.x == 0) {
    float* tail = out + (size_t)n_img * n_cell * C;
    for (int i = threadIdx.x; i < kTailCells(Wf) * C; i += 256) tail[i] = 0.f;
  }
}

#ifdef V3D_PHASE_TIMING
// developer build only: per-phase cycle counters, see costreg.hip
constexpr int kPhaseSlots = 1 << 16;
__device__ unsigned long long g_psv_phase[8 * kPhaseSlots];
#define PHASE_DECL                                  \
  long long ph_t = __builtin_readcyclecounter();    \
  long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_MARK(i)                                   \
  do {                                                  \
    long long t_ = __builtin_readcyclecounter();        \
    ph_acc[i] += t_ - ph_t;                             \
    ph_t = t_;                                          \
  } while (0)
#define PHASE_FLUSH                                                                                       \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < kPhaseSlots)                                                     \
      for (int i_ = 0; i_ < 8; ++i_) g_psv_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_];   \
  } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_FLUSH
#endif

// Per-image camera block: [0..8] K^-1, [9..17] R, [18..20] t (used when the image is a reference view) and
// [24..35] P = K [R|t] (used when it is a source view).
constexpr int kCamStride = 36;

__global__ void cam_setup_kernel(const float* __restrict__ K, const float* __restrict__ R,
                                 const float* __restrict__ t, float* __restrict__ camp, int n_img) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n_img) return;
  const float* Kp = K + img * 9;
  const float* Rp = R + img * 9;
  const float* tp = t + img * 3;
  float* o = camp + img * kCamStride;
  // K^-1 in fp64 (adjugate / determinant), rounded to f32 (utils.py:103 torch.inverse)
  double a = Kp[0], bb = Kp[1], c = Kp[2], d = Kp[3], e = Kp[4], f = Kp[5], g = Kp[6], hh = Kp[7], i = Kp[8];
  double det = a * (e * i - f * hh) - bb * (d * i - f * g) + c * (d * hh - e * g);
  double id = 1.0 / det;
  o[0] = (float)((e * i - f * hh) * id);
  o[1] = (float)((c * hh - bb * i) * id);
  o[2] = (float)((bb * f - c * e) * id);
  o[3] = (float)((f * g - d * i) * id);
  o[4] = (float)((a * i - c * g) * id);
  o[5] = (float)((c * d - a * f) * id);
  o[6] = (float)((d * hh - e * g) * id);
  o[7] = (float)((bb * g - a * hh) * id);
  o[8] = (float)((a * e - bb * d) * id);
  for (int k = 0; k < 9; ++k) o[9 + k] = Rp[k];
  for (int k = 0; k < 3; ++k) o[18 + k] = tp[k];
  // projection matrix P = K [R|t] (mvsnet.py:196-197).  torch.bmm evaluates this small batched product with rounded
  // products and sequential additions (no FMA) -- unlike the large K^-1 p / R^T c / P X products, which are FMA chains
  // in k order -- so contraction is switched off here: the sample coordinates then reproduce the reference's bit for bit
  // (scripts/coord_order_probe.py)
  for (int r = 0; r < 3; ++r)
    for (int j = 0; j < 4; ++j) {
      const float b0 = j < 3 ? Rp[0 * 3 + j] : tp[0], b1 = j < 3 ? Rp[1 * 3 + j] : tp[1], b2 = j < 3 ? Rp[2 * 3 + j] : tp[2];
      o[24 + r * 4 + j] = v3d::add_rn(v3d::add_rn(v3d::mul_rn(Kp[r * 3 + 0], b0), v3d::mul_rn(Kp[r * 3 + 1], b1)),
                                    v3d::mul_rn(Kp[r * 3 + 2], b2));
    }
}

struct TapInfo {      // one (pixel, plane, edge) sample, 32 bytes
  float w00, w01, w10, w11;      // nw, ne, sw, se weights: the reference's products (x1 - ix)(y1 - iy), ...
  unsigned b00;                  // byte offset of the nw cell (channel 0) in the bordered featT; ne = +CB, sw = +row, se = +row + CB
  unsigned pad[3];
};

// Bilinear taps of one sample (F.grid_sample, align_corners=True, zeros padding; mvsnet.py:209-211) on the BORDERED
// channel-last feature maps (transpose_bordered_kernel).  The sample position is clamped to [-1, Wf] x [-1, Hf] first: inside
// that range nothing changes -- a tap outside the image is a zero cell of the border and contributes exactly 0 with its true
// weight, as in the reference, which zeroes the tap's value; outside it every tap of the true sample is out of range (result
// 0), and the clamped sample sits on the border with weight 1 on zero cells and weight 0 on the others: 0 as well.  NaN
// positions clamp to -1 (fmaxf returns the number).  So there is no validity test, no per-tap clamp and one offset per
// sample instead of four.  CB = bytes per cell (4 C); first_cell = index of cell (-1, -1) of the source image.
template <unsigned CB>
__device__ __forceinline__ TapInfo make_taps(float ix, float iy, int Wf, int Hf, int first_cell) {
  ix = fminf(fmaxf(ix, -1.f), (float)Wf);
  iy = fminf(fmaxf(iy, -1.f), (float)Hf);
  const float x0 = floorf(ix), y0 = floorf(iy);
  const float x1 = x0 + 1.f, y1 = y0 + 1.f;
  const float wx0 = x1 - ix, wx1 = ix - x0, wy0 = y1 - iy, wy1 = iy - y0;
  TapInfo ti;
  ti.w00 = v3d::mul_rn(wx0, wy0); ti.w01 = v3d::mul_rn(wx1, wy0);
  ti.w10 = v3d::mul_rn(wx0, wy1); ti.w11 = v3d::mul_rn(wx1, wy1);
  ti.b00 = (unsigned)(first_cell + ((int)y0 + 1) * (Wf + 2) + (int)x0 + 1) * CB;
  return ti;
}

// SPLIT: write the variance as the regulariser's first layer consumes it (costreg.hip, conv0_bf16x2_kernel):
// every value split x = hi + lo into two bf16, channel-last in 16-byte slots of 8 channels,
// [n_ref][4 channel groups][hi, lo][D][h][w][8] -- the same 4 bytes per value as fp32, and bit-identical to
// splitting the fp32 volume later, but conv0 then stages tiles with plain 16-byte copies.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned psv_bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// two fp32 -> two bf16 (round to nearest even) packed in one dword, a in the low half: one v_cvt_pk_bf16_f32
typedef __bf16 psv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, psv_bf16x2));
}

// THREADS = 64: one wave per workgroup (16 pixels x 4 planes).  The projection, gather and store phases of a
// workgroup are separated by barriers; with single-wave workgroups the barriers are free and the 24 resident
// waves of a CU drift apart, so the L1 (the binding resource, busy only during the gather phase) always has some
// wave gathering.
template <int C, bool SPLIT, int THREADS>
__global__ __launch_bounds__(THREADS) void psv_variance_kernel(PsvParams p) {
  constexpr int kThreads = THREADS, kPix = THREADS / 4;
  constexpr int LP = C / 4;               // lanes per pixel
  constexpr int PPP = kThreads / LP;      // pixels per phase-2 pass
  constexpr int NPASS = kPix / PPP;
  static_assert(kPix % PPP == 0, "tile");

  __shared__ TapInfo s_tap[kMaxE * kPix];
  __shared__ __attribute__((aligned(16))) float s_out[C][kPix + 1];
  __shared__ float s_ref[24];             // Kinv(9) R(9) t(3)
  __shared__ float s_P[kMaxE][12];

  const int tid = threadIdx.x;
  const int n_dchunk = (p.D + kDB - 1) / kDB;
  // every XCD takes a contiguous run of reference views, so its L2 holds the source feature maps of one
  // reference (~E x 400 KB) instead of all 8 dies streaming the sources of every reference in flight
  int b = v3d::xcd_contiguous_block();
  const int ptile = b % p.n_ptile; b /= p.n_ptile;
  const int dchunk = b % n_dchunk;
  const int r = b / n_dchunk;
  const int P = p.h * p.w;
  const int e_begin = p.edge_ofs[r], e_end = p.edge_ofs[r + 1];
  const int ne = e_end - e_begin;
  const int ref = p.ref_img[r];

  // K^-1 (9), R (9), t (3) of the reference view, prepared by cam_setup_kernel
  if (tid < 21) s_ref[tid] = p.camp[ref * kCamStride + tid];

  // this thread's phase-2 role
  const int cg = tid % LP;
  const int pix_in_pass = tid / LP;

  // pixel coordinates of this thread's phase-1 pixel (same pixel for every pair it handles)
  const int px1 = tid % kPix;
  const int gp1 = ptile * kPix + px1;
  float xf = 0.f, yf = 0.f;
  {
    int gy = gp1 / p.w, gx = gp1 % p.w;
    // numpy.linspace(0, W-1, w, dtype=float32): float64 arithmetic, last sample = stop
    xf = (p.w > 1 && gx == p.w - 1) ? (float)(p.W - 1) : (float)((double)gx * p.x_step);
    yf = (p.h > 1 && gy == p.h - 1) ? (float)(p.H - 1) : (float)((double)gy * p.y_step);
  }
  const float Wm1 = (float)(p.W - 1), Hm1 = (float)(p.H - 1);
  const float rWm1 = (float)(1.0 / (double)(p.W - 1)), rHm1 = (float)(1.0 / (double)(p.H - 1));
  const float Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);

  PHASE_DECL;
  for (int dd = 0; dd < kDB; ++dd) {
    const int d = dchunk * kDB + dd;
    if (d >= p.D) break;
    const float z = (d == p.D - 1 && p.D > 1) ? (float)p.z_end
                                               : (float)(p.z_start + (double)d * p.z_step);
    float acc_s[NPASS][4], acc_q[NPASS][4];
#pragma unroll
    for (int a = 0; a < NPASS; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc_s[a][k] = acc_q[a][k] = 0.f;

    for (int ec = 0; ec < ne; ec += kMaxE) {
      const int nec = min(kMaxE, ne - ec);
      __syncthreads();   // previous users of s_tap / s_P / s_out are done; s_ref visible
      PHASE_MARK(0);
      // projection matrices P = K [R|t] of this chunk's source views (mvsnet.py:196-197); with a single chunk
      // of edges (the usual case) they are the same for every plane of the workgroup: computed once
      if (dd == 0 || ne > kMaxE) {
        for (int i = tid; i < nec * 12; i += kThreads)
          s_P[i / 12][i % 12] = p.camp[p.edge_src[e_begin + ec + i / 12] * kCamStride + 24 + i % 12];
      }
      __syncthreads();
      PHASE_MARK(1);
      // ---- phase 1: project (pixel, edge) pairs ------------------------------------------
      {
        float X, Y, Z;
        v3d::world_point(s_ref, xf, yf, z, X, Y, Z);
        for (int e = tid / kPix; e < nec; e += kThreads / kPix) {
          float ix, iy;
          v3d::sample_position(s_P[e], X, Y, Z, Wm1, rWm1, Hm1, rHm1, Wfm1, Hfm1, ix, iy);
          const TapInfo ti = make_taps<4 * C>(ix, iy, p.Wf, p.Hf, p.edge_src[e_begin + ec + e] * (p.Hf + 2) * (p.Wf + 2));
          s_tap[e * kPix + px1] = ti;
        }
      }
      PHASE_MARK(2);
      __syncthreads();
      PHASE_MARK(3);
      // ---- phase 2: gather + accumulate ----------------------------------------------------
#pragma unroll
      for (int a = 0; a < NPASS; ++a) {
        const int px = a * PPP + pix_in_pass;
        for (int e = 0; e < nec; ++e) {
          const TapInfo ti = s_tap[e * kPix + px];
          float4 s;
          {
            // uniform base + 32-bit byte offset: the load takes the address as SGPR pair + VGPR offset
            const char* fb = reinterpret_cast<const char*>(p.featT);
            const unsigned cgb = cg * 16, rowb = (unsigned)(p.Wf + 2) * (4 * C);
            const float4 v00 = *reinterpret_cast<const float4*>(fb + (size_t)(ti.b00 + cgb));
            const float4 v01 = *reinterpret_cast<const float4*>(fb + (size_t)(ti.b00 + cgb) + 4 * C);
            const float4 v10 = *reinterpret_cast<const float4*>(fb + (size_t)(ti.b00 + rowb + cgb));
            const float4 v11 = *reinterpret_cast<const float4*>(fb + (size_t)(ti.b00 + rowb + cgb) + 4 * C);
            // F.grid_sample's tap order as an FMA chain: ((v00 w00 + v01 w01) + v10 w10) + v11 w11 -- the plane-reuse kernel
            // below uses the same sequence, so the variants agree bit for bit
            s.x = v3d::mul_rn(v00.x, ti.w00); s.y = v3d::mul_rn(v00.y, ti.w00);
            s.z = v3d::mul_rn(v00.z, ti.w00); s.w = v3d::mul_rn(v00.w, ti.w00);
            s.x = __builtin_fmaf(v01.x, ti.w01, s.x); s.y = __builtin_fmaf(v01.y, ti.w01, s.y);
            s.z = __builtin_fmaf(v01.z, ti.w01, s.z); s.w = __builtin_fmaf(v01.w, ti.w01, s.w);
            s.x = __builtin_fmaf(v10.x, ti.w10, s.x); s.y = __builtin_fmaf(v10.y, ti.w10, s.y);
            s.z = __builtin_fmaf(v10.z, ti.w10, s.z); s.w = __builtin_fmaf(v10.w, ti.w10, s.w);
            s.x = __builtin_fmaf(v11.x, ti.w11, s.x); s.y = __builtin_fmaf(v11.y, ti.w11, s.y);
            s.z = __builtin_fmaf(v11.z, ti.w11, s.z); s.w = __builtin_fmaf(v11.w, ti.w11, s.w);
          }
          acc_s[a][0] = v3d::add_rn(acc_s[a][0], s.x); acc_s[a][1] = v3d::add_rn(acc_s[a][1], s.y);
          acc_s[a][2] = v3d::add_rn(acc_s[a][2], s.z); acc_s[a][3] = v3d::add_rn(acc_s[a][3], s.w);
          acc_q[a][0] = __builtin_fmaf(s.x, s.x, acc_q[a][0]); acc_q[a][1] = __builtin_fmaf(s.y, s.y, acc_q[a][1]);
          acc_q[a][2] = __builtin_fmaf(s.z, s.z, acc_q[a][2]); acc_q[a][3] = __builtin_fmaf(s.w, s.w, acc_q[a][3]);
        }
      }
    }
    PHASE_MARK(4);
    // ---- variance, transpose through LDS, coalesced store -------------------------------------
    const float cnt = (float)max(ne, 1);        // torch_scatter mean: sum / clamp(count, 1)
    if constexpr (SPLIT) {
      static_assert(!SPLIT || C == 32, "split layout is defined for 32 channels");
      // s_out reused as [8 groups][kPix + 1] 16-byte slots (the +1 keeps the 4 channel groups off one bank)
      u32x2* const s_sp = reinterpret_cast<u32x2*>(&s_out[0][0]);
      const int chunk = cg >> 1, half = cg & 1;
#pragma unroll
      for (int a = 0; a < NPASS; ++a) {
        const int px = a * PPP + pix_in_pass;
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float avg = acc_s[a][k] / cnt;
          const float avg_sq = acc_q[a][k] / cnt;
          const float v = v3d::sub_rn(avg_sq, v3d::mul_rn(avg, avg));            // mvsnet.py:216
          h[k] = psv_bf16_rne(v);
          l[k] = psv_bf16_rne(v - __uint_as_float(h[k] << 16));
        }
        s_sp[(((chunk * 2 + 0) * (kPix + 1)) + px) * 2 + half] = (u32x2){h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        s_sp[(((chunk * 2 + 1) * (kPix + 1)) + px) * 2 + half] = (u32x2){l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
      }
      __syncthreads();
      u32x4* const out = reinterpret_cast<u32x4*>(p.var);
      const u32x4* const s_q = reinterpret_cast<const u32x4*>(s_sp);
      for (int i = tid; i < 8 * kPix; i += kThreads) {
        const int g = i / kPix, px = i % kPix;
        const int gp = ptile * kPix + px;
        if (gp < P) __builtin_nontemporal_store(s_q[g * (kPix + 1) + px], &out[(((size_t)r * 8 + g) * p.D + d) * P + gp]);
      }
    } else {
#pragma unroll
      for (int a = 0; a < NPASS; ++a) {
        const int px = a * PPP + pix_in_pass;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float avg = acc_s[a][k] / cnt;
          float avg_sq = acc_q[a][k] / cnt;
          s_out[cg * 4 + k][px] = v3d::sub_rn(avg_sq, v3d::mul_rn(avg, avg));   // mvsnet.py:216
        }
      }
      __syncthreads();
      {
        const int px = tid % kPix;
        const int gp = ptile * kPix + px;
        if (gp < P) {
          for (int c = tid / kPix; c < C; c += kThreads / kPix)
            __builtin_nontemporal_store(s_out[c][px], &p.var[(((size_t)r * C + c) * p.D + d) * P + gp]);
        }
      }
    }
    PHASE_MARK(5);
  }
  PHASE_FLUSH;
}


// ---------------------------------------------------------------------------------------------------
// Plane-reuse variant (C == 32), the default.  The gather kernel above is bound by L1 bytes: 4 taps x 128 B per
// (pixel, plane, source view).  But the samples of one pixel on consecutive planes walk along its epipolar line in
// sub-pixel steps on most of the depth range, so their 2x2 footprints coincide (cfg2: the 16 taps of 4 consecutive
// planes touch 5.1 distinct cells on average).  Here a single-wave workgroup owns 8 pixels x kDB planes; the 8
// lanes of a pixel keep the footprint of the previous plane in registers and reload it only when the footprint
// of the next plane differs, edge by edge.  Arithmetic and accumulation order per (pixel, plane) are those of
// the gather kernel (bit-identical output).
//
// What bounds it (round-2 measurements, cfg2, 64 views): no single pipe.  A step = one (plane, edge) for the
// wave's 8 pixels; a footprint reload is four 1 KB wave loads, issued when ANY of the 8 pixels changed its footprint, with the
// other pixels' lanes masked: 1.8 wave loads per step at 8 planes per wave (scripts/sim_footprint_reloads.py), ~29 of the ~48
// cycles a CU spends per step; VALU ~70 % busy, LDS ~35 %, HBM traffic 1.08x algorithmic.  Measured: 18 % fewer VALU
// instructions in the loop (bordered maps, one offset per sample): no change; 8 planes per wave instead of 4 (2.1 -> 1.8
// loads per step): -7 %, although only 4 waves per SIMD fit; 2 planes per wave at 8 waves per SIMD: +19 %; partial reloads,
// a persistent tile walk, 4x2 pixel blocks: equal or slower.
constexpr int kRPix = 8;      // pixels per wave
#ifndef V3D_PSV_RDB
#define V3D_PSV_RDB 8
#endif
constexpr int kRDB = V3D_PSV_RDB;                 // depth planes per wave
constexpr int kRE = 64 / (kRDB * kRPix);          // edges per phase-1 pass: kRE x kRDB x kRPix = 64 (pixel, plane, edge) items, one per lane

#ifndef V3D_PSV_WAVES
#define V3D_PSV_WAVES 4
#endif
#ifndef V3D_PSV_ABLATE
#define V3D_PSV_ABLATE 0     // developer ablations of the reuse kernel (scripts/ab_build.sh): 1 no blend, 2 no reloads, 3 no projection, 4 no store
#endif
// Waves of one workgroup only meet in the fp32 epilogue: everything else a wave exchanges through LDS is its own (LDS
// operations of a wave complete in order, so a fence that keeps the compiler from reordering them is enough).
template <int WPB>
__device__ __forceinline__ void psv_wave_sync() {
  if constexpr (WPB == 1) __syncthreads();
  else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
}

// SPLIT = true: single-wave workgroups, the variance leaves in the regulariser's split format (16-byte slots, 128-byte runs per
// wave).  SPLIT = false (the public fp32 tensor [n_ref, C, D, h, w]): FOUR waves = four consecutive 8-pixel tiles per workgroup;
// they work independently and only share the output staging, so that a (channel, plane) row leaves as 32 consecutive pixels =
// 128-byte runs.  With one wave per workgroup the rows were 32-byte runs: 3.95 GB written for a 2.47 GB volume, 3.6 ms per 64
// views against 1.7 ms for the split variant.
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 64 : 256, V3D_PSV_WAVES) void psv_variance_reuse_kernel(PsvParams p) {
  constexpr int C = 32, WPB = SPLIT ? 1 : 4;
  constexpr int kOutPl = 4;               // planes staged per round of the fp32 epilogue
  static_assert(SPLIT || kRDB % kOutPl == 0, "fp32 epilogue stages 4 planes per round");
  __shared__ TapInfo s_tap_[WPB][kRE * kRDB * kRPix];
  __shared__ __attribute__((aligned(16))) float s_out[SPLIT ? 1 : kOutPl][SPLIT ? 1 : C][SPLIT ? 1 : WPB * kRPix + 1];
  __shared__ float s_ref_[WPB][24];
  __shared__ float s_P_[WPB][kMaxE][12];  // projection matrices of up to kMaxE consecutive edges
  __shared__ int s_base_[WPB][kMaxE];     // first feature cell of their source images

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  TapInfo* const s_tap = s_tap_[wv];
  float* const s_ref = s_ref_[wv];
  float (*const s_P)[12] = s_P_[wv];
  int* const s_base = s_base_[wv];
  const int n_dchunk = (p.D + kRDB - 1) / kRDB;
  int b = v3d::xcd_contiguous_block();
  // plane chunks fastest: the waves resident on an XCD sweep all planes of a few pixel tiles, i.e. short epipolar segments of
  // the source maps, instead of one plane chunk of the whole image (= every source map entirely, more than the 4 MB L2)
  const int dchunk = b % n_dchunk; b /= n_dchunk;
  const int ptile = (b % p.n_ptile) * WPB + wv;      // p.n_ptile counts groups of WPB tiles
  const int r = b / p.n_ptile;
  const int P = p.h * p.w;
  const int e_begin = p.edge_ofs[r], e_end = p.edge_ofs[r + 1];
  const int ne = e_end - e_begin;
  const int ref = p.ref_img[r];
  if (lane < 21) s_ref[lane] = p.camp[ref * kCamStride + lane];
  psv_wave_sync<WPB>();

  // phase-1 role: lane = (edge slot, plane, pixel); the world point of (pixel, plane) is the same for every edge
  const int e1 = lane / (kRDB * kRPix), pl1 = (lane >> 3) & (kRDB - 1), px1 = lane & 7;
  const int gp1 = ptile * kRPix + px1;
  const int d1 = dchunk * kRDB + pl1;
  float X, Y, Z;
  {
    const int gy = gp1 / p.w, gx = gp1 % p.w;
    // numpy.linspace(0, W-1, w, dtype=float32): float64 arithmetic, last sample = stop
    const float xf = (p.w > 1 && gx == p.w - 1) ? (float)(p.W - 1) : (float)((double)gx * p.x_step);
    const float yf = (p.h > 1 && gy == p.h - 1) ? (float)(p.H - 1) : (float)((double)gy * p.y_step);
    const float z = (d1 == p.D - 1 && p.D > 1) ? (float)p.z_end : (float)(p.z_start + (double)d1 * p.z_step);
    v3d::world_point(s_ref, xf, yf, z, X, Y, Z);
  }
  const bool live1 = gp1 < P && d1 < p.D;
  const float Wm1 = (float)(p.W - 1), Hm1 = (float)(p.H - 1);
  // divisions by the wave-uniform image extents go through their correctly rounded reciprocals (v3d::div_uniform)
  const float rWm1 = (float)(1.0 / (double)(p.W - 1)), rHm1 = (float)(1.0 / (double)(p.H - 1));
  const float Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);

  // gather role: 8 lanes x float4 per pixel
  const int gpx = lane >> 3;
  const unsigned cgb = (lane & 7) * 16;
  const unsigned rowb = (unsigned)(p.Wf + 2) * (4 * C) + cgb;      // sw cell of a footprint = nw + one bordered row
  const char* const fb = reinterpret_cast<const char*>(p.featT);
  f32x4 acc_s[kRDB], acc_q[kRDB];
#pragma unroll
  for (int k = 0; k < kRDB; ++k) acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  PHASE_DECL;
  for (int ec = 0; ec < ne; ec += kRE) {
    const int nec = min(kRE, ne - ec);
    psv_wave_sync<WPB>();                  // the previous pass is done with s_tap / s_P
    PHASE_MARK(0);
    if (ec % kMaxE == 0) {
      const int nload = min(kMaxE, ne - ec) * 12;
#pragma unroll 1
      for (int i = lane; i < nload; i += 64) {
        const int src = p.edge_src[e_begin + ec + i / 12];
        s_P[i / 12][i % 12] = p.camp[src * kCamStride + 24 + i % 12];
        if (i % 12 == 0) s_base[i / 12] = src * (p.Hf + 2) * (p.Wf + 2);      // cell (-1, -1) of the bordered map
      }
      psv_wave_sync<WPB>();
    }
#if V3D_PSV_ABLATE == 3      // developer ablation: no projection (tap records written once)
    if (ec == 0)
#endif
    if (e1 < nec) {
      float ix, iy;
      v3d::sample_position(s_P[ec % kMaxE + e1], X, Y, Z, Wm1, rWm1, Hm1, rHm1, Wfm1, Hfm1, ix, iy);
      s_tap[(e1 * kRDB + pl1) * kRPix + px1] = make_taps<4 * C>(ix, iy, p.Wf, p.Hf, s_base[ec % kMaxE + e1]);
    }
    psv_wave_sync<WPB>();
    PHASE_MARK(1);
    for (int e = 0; e < nec; ++e) {
      unsigned c00 = ~0u;                // footprint held in t00..t11 (its nw cell pins all four)
      // deliberately not initialised: ~0 matches no offset, so the first plane always loads them
      f32x4 t00, t01, t10, t11;
#pragma unroll
      for (int pl = 0; pl < kRDB; ++pl) {
        const TapInfo ti = s_tap[(e * kRDB + pl) * kRPix + gpx];
#if V3D_PSV_ABLATE == 2      // developer ablation: footprint loaded once per edge only
        if (pl == 0) {
#else
        if (ti.b00 != c00) {
#endif
          // (Reloading only the two new cells when the footprint moved by one cell -- register moves for the other two -- was
          // slower, 2.02 vs 1.84 ms: a step then issues two loads per direction the wave's pixels moved in, and the
          // vector-memory path is charged per wave instruction, not per active lane.)
          t00 = *reinterpret_cast<const f32x4*>(fb + (size_t)(ti.b00 + cgb));
          t01 = *reinterpret_cast<const f32x4*>(fb + (size_t)(ti.b00 + cgb) + 4 * C);
          t10 = *reinterpret_cast<const f32x4*>(fb + (size_t)(ti.b00 + rowb));
          t11 = *reinterpret_cast<const f32x4*>(fb + (size_t)(ti.b00 + rowb) + 4 * C);
          c00 = ti.b00;
        }
        // F.grid_sample's tap order as an FMA chain (same sequence as the gather kernel); no validity branch: taps outside
        // the source image read zero cells of the border
#if V3D_PSV_ABLATE == 1      // developer ablation: no blend arithmetic (loads and tap reads kept alive)
        asm volatile("" : : "v"(t00), "v"(t01), "v"(t10), "v"(t11), "v"(ti.w00), "v"(ti.w01), "v"(ti.w10), "v"(ti.w11));
#else
        f32x4 sv = t00 * ti.w00;
        sv = __builtin_elementwise_fma(t01, (f32x4){ti.w01, ti.w01, ti.w01, ti.w01}, sv);
        sv = __builtin_elementwise_fma(t10, (f32x4){ti.w10, ti.w10, ti.w10, ti.w10}, sv);
        sv = __builtin_elementwise_fma(t11, (f32x4){ti.w11, ti.w11, ti.w11, ti.w11}, sv);
        acc_s[pl] += sv;
        acc_q[pl] = __builtin_elementwise_fma(sv, sv, acc_q[pl]);
#endif
#ifndef V3D_PSV_NOSERIAL
        // Finish this plane before the next one starts: the empty statement pins the accumulators here and, as a memory
        // barrier, keeps the next plane's tap-record read and footprint loads below it.  Without it the compiler issues the
        // (conditional) loads of all four planes first, each into its own 16 registers: 136 VGPRs, 3 waves per SIMD.
        asm volatile("" : "+v"(acc_s[pl]), "+v"(acc_q[pl]) : : "memory");
#endif
      }
    }
    PHASE_MARK(4);
  }

  PHASE_MARK(2);
  // ---- variance -> LDS -> stores ------------------------------------------------------------------------------
  psv_wave_sync<WPB>();
  const float cnt = (float)max(ne, 1);        // torch_scatter mean: sum / clamp(count, 1)
  // x / cnt is an IEEE division (~10 instructions); for a power-of-two count (8 views in the headline configuration)
  // x * (1 / cnt) is the same number exactly.  Wave-uniform choice.
  const bool cnt_pow2 = (max(ne, 1) & (max(ne, 1) - 1)) == 0;
  const float cnt_inv = 1.f / cnt;
#ifdef V3D_PSV_EXACT_DIV
  auto mean = [&](float x) __attribute__((always_inline)) { return x / cnt; };
#else
  auto mean = [&](float x) __attribute__((always_inline)) { return cnt_pow2 ? x * cnt_inv : x / cnt; };
#endif
  const int cg = lane & 7;
  if constexpr (SPLIT) {
    // Lane (pixel gpx, channel quad cg) holds channels 4 cg .. 4 cg + 3 = one 8-byte half of the 16-byte hi slot and of the
    // lo slot of channel group cg / 2.  Lane pairs swap halves (one DPP move per dword): the even lane assembles and stores
    // the whole hi slot, the odd lane the whole lo slot -- no LDS round trip; the 8 pixels of a wave make 128-byte runs.
    const int chunk = cg >> 1, half = cg & 1;
    u32x4* const out = reinterpret_cast<u32x4*>(p.var);
    const int gp = ptile * kRPix + gpx;
#pragma unroll
    for (int pl = 0; pl < kRDB; ++pl) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float avg = mean(acc_s[pl][k]);
        const float avg_sq = mean(acc_q[pl][k]);
        v[k] = v3d::sub_rn(avg_sq, v3d::mul_rn(avg, avg));                    // mvsnet.py:216
      }
      // x = hi + lo: hi = RNE_bf16(x), lo = RNE_bf16(x - hi); v_cvt_pk_bf16_f32 rounds to nearest even in hardware
      const unsigned h01 = pack_bf16x2(v[0], v[1]), h23 = pack_bf16x2(v[2], v[3]);
      const unsigned l01 = pack_bf16x2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
      const unsigned l23 = pack_bf16x2(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));
      const unsigned s0 = half ? h01 : l01, s1 = half ? h23 : l23;             // what the partner lane stores
      const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
      const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xf, 0xf, true);
      const u32x4 slot = half ? (u32x4){r0, r1, l01, l23} : (u32x4){h01, h23, r0, r1};
      const int d = dchunk * kRDB + pl;
      if (gp < P && d < p.D && (V3D_PSV_ABLATE != 4 || slot[0] == 0x12345u))
#ifdef V3D_PSV_PLAIN_STORE      // developer A/B
        out[(((size_t)r * 8 + chunk * 2 + half) * p.D + d) * P + gp] = slot;
#else
        __builtin_nontemporal_store(slot, &out[(((size_t)r * 8 + chunk * 2 + half) * p.D + d) * P + gp]);
#endif
    }
  } else {
    // fp32 tensor: the four waves park kOutPl planes of their 8 pixels side by side ([plane][channel][32 pixels]) and the
    // workgroup stores every (channel, plane) row as one 128-byte run
    constexpr int NPX = WPB * kRPix;
    const int gp0 = (ptile - wv) * kRPix;             // first pixel of the workgroup
#pragma unroll
    for (int round = 0; round < kRDB / kOutPl; ++round) {
      if (round) __syncthreads();                      // the previous round's rows are stored
#pragma unroll
      for (int q = 0; q < kOutPl; ++q) {
        const int pl = round * kOutPl + q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float avg = mean(acc_s[pl][k]);
          const float avg_sq = mean(acc_q[pl][k]);
          s_out[q][cg * 4 + k][wv * kRPix + gpx] = v3d::sub_rn(avg_sq, v3d::mul_rn(avg, avg));   // mvsnet.py:216
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < kOutPl * C * NPX; i += 64 * WPB) {
        const int q = i / (C * NPX), c = (i / NPX) % C, px = i % NPX;
        const int gp = gp0 + px, d = dchunk * kRDB + round * kOutPl + q;
        if (gp < P && d < p.D) __builtin_nontemporal_store(s_out[q][c][px], &p.var[(((size_t)r * C + c) * p.D + d) * P + gp]);
      }
    }
  }
  PHASE_MARK(3);
  PHASE_FLUSH;
}

// ---------------------------------------------------------------------------------------------------
// Window variant of the plane-reuse kernel (C == 32, 8 planes per wave), the default since round 3.
//
// Round-3 measurements behind it (scripts/micro/ta_mask.hip): a 16-byte-per-lane gather costs the CU's
// vector-memory path 7.2 ns per WAVE INSTRUCTION -- whatever the exec mask (8 or 64 active lanes) and whether a lane moves 8 or
// 16 bytes -- and the reuse kernel issues 1.8 of them per (plane, edge) step: 0.98 ms of its 1.65 ms per 64 cfg2 views are
// that path's busy time, queued behind each other (the ~1 300-cycle reload latency a wave sees).  Its reloads are mostly
// waste: a wave's 8 pixels x 8 planes x 1 edge touch 16 DISTINCT cells on average (the 2x2 footprints of neighbouring
// pixels and of consecutive planes overlap), the kernel loads 115 (forced reload of every pixel on the first plane, all four
// cells again when a footprint moves by one cell, masked lanes at full price).
//
// Here the wave takes the box spanned by the four corner samples of its 64 footprints per edge, cut to 16 x 4 cells (the whole
// box in 95 % of the passes at cfg2, 97 % at cfg5), and copies it into LDS with global_load_lds_dwordx4 -- one instruction
// per 8 cells of a row, no staging registers, 0.53 instructions per step instead of 1.8 -- and the footprints are read from
// there (ds_read_b128, a quarter of the L1 path's price per KB, tens instead of hundreds of cycles of latency).  Whether any
// pixel changes its footprint on plane k is known after the projection (one ballot): the reload branch is scalar and all
// lanes reload together.  A sample whose footprint lies outside the window (near planes with long epipolar slides, exotic
// camera pairs, the few tiles whose corners are not extreme) carries a flag in its tap word (the sign bit) and reads its four cells from
// featT as the reuse kernel does, inside the same step.  Arithmetic and accumulation order are those of the other two
// kernels: bit-identical output.
constexpr int kWinCols = 16, kWinRows = 4;                 // window: 16 x 4 cells x 128 B = 8 KB per wave
typedef float psv_f32x2 __attribute__((ext_vector_type(2)));

// scalar min / max of wave-uniform values (the compiler picks v_min3 / v_max3 + v_readfirstlane for these otherwise)
__device__ __forceinline__ int psv_smin(int a, int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int psv_smax(int a, int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }

#ifndef V3D_PSVW_ABLATE
#define V3D_PSVW_ABLATE 0    // developer ablations of the window kernel: 1 no blend, 2 no footprint reads, 3 no window copy, 5 no store, 6 no out-of-window path
#endif
// CL8 (with SPLIT's geometry: one wave per workgroup, direct stores): the volume leaves as fp32 in the channel-last layout
// of the exact-fp32 depth-march conv0 (conv0z.hip) -- [n_ref][4 channel groups][2 halves][D][h][w] 16-byte slots of 4 floats
// (include/v3d.h, v3d_psv_variance_cl8): the split layout's addressing with the values themselves instead of bf16 pairs.
template <bool SPLIT, bool CL8 = false>
__global__ __launch_bounds__(SPLIT ? 64 : 256, V3D_PSV_WAVES) void psv_variance_window_kernel(PsvParams p) {
  static_assert(SPLIT || !CL8, "the channel-last fp32 output uses the single-wave geometry");
  constexpr int C = 32, WPB = SPLIT ? 1 : 4;
  constexpr unsigned CB = 4 * C;                            // bytes per cell
  constexpr int kOutPl = 4;                                 // planes staged per round of the fp32 epilogue
  static_assert(kWinCols == 16 && kRDB == 8 && kRE == 1, "the window kernel projects one edge per pass: 8 planes x 8 pixels = 64 lanes");
  __shared__ __attribute__((aligned(16))) float s_win_[WPB][kWinRows * kWinCols * C];
  __shared__ __attribute__((aligned(16))) f32x4 s_w_[WPB][kRDB * kRPix];      // nw, ne, sw, se weights per (plane, pixel)
  __shared__ unsigned s_slot_[WPB][kRDB * kRPix];           // byte offset of the nw cell: in the window / in featT
  __shared__ float s_ref_[WPB][24];
  __shared__ float s_P_[WPB][kMaxE][12];
  __shared__ int s_base_[WPB][kMaxE];
  // fp32 epilogue: [kOutPl][C][WPB * kRPix + 1] floats = 16.5 KB, parked on the four (by then dead) windows
  static_assert(SPLIT || kOutPl * C * (WPB * kRPix + 1) <= WPB * kWinRows * kWinCols * C, "epilogue staging fits the windows");

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const s_win = s_win_[wv];
  f32x4* const s_w = s_w_[wv];
  unsigned* const s_slot = s_slot_[wv];
  float* const s_ref = s_ref_[wv];
  float (*const s_P)[12] = s_P_[wv];
  int* const s_base = s_base_[wv];
  const int n_dchunk = (p.D + kRDB - 1) / kRDB;
  const unsigned b = (unsigned)v3d::xcd_contiguous_block();
  const unsigned b1 = v3d::udiv_magic(b, (unsigned)n_dchunk, p.m_dchunk), b2 = v3d::udiv_magic(b1, (unsigned)p.n_ptile, p.m_ptile);
  const int dchunk = (int)(b - b1 * (unsigned)n_dchunk);     // plane chunks fastest (see the reuse kernel)
  const int ptile = (int)(b1 - b2 * (unsigned)p.n_ptile) * WPB + wv;
  const int r = (int)b2;
  const int P = p.h * p.w;
  const int e_begin = p.edge_ofs[r], e_end = p.edge_ofs[r + 1];
  const int ne = e_end - e_begin;
  const int ref = p.ref_img[r];
  if (lane < 21) s_ref[lane] = p.camp[ref * kCamStride + lane];
  psv_wave_sync<WPB>();

  // projection role: lane = (plane, pixel)
  const int pl1 = lane >> 3, px1 = lane & 7;
  const int gp1 = ptile * kRPix + px1;
  const int d1 = dchunk * kRDB + pl1;
  float X, Y, Z;
  {
    // row / column of the lane's pixel: the wave's first pixel is divided once (wave-uniform), the lane adds its offset
    // (rows of at least kRPix pixels wrap at most once)
    int gy, gx;
    if (p.w >= kRPix) {
      const unsigned g0 = (unsigned)(ptile * kRPix), gy0 = v3d::udiv_magic(g0, (unsigned)p.w, p.m_w);
      gx = (int)(g0 - gy0 * (unsigned)p.w) + px1;
      gy = (int)gy0;
      if (gx >= p.w) { gx -= p.w; ++gy; }
    } else {
      gy = gp1 / p.w; gx = gp1 % p.w;
    }
    const float xf = (p.w > 1 && gx == p.w - 1) ? (float)(p.W - 1) : (float)((double)gx * p.x_step);
    const float yf = (p.h > 1 && gy == p.h - 1) ? (float)(p.H - 1) : (float)((double)gy * p.y_step);
    const float z = (d1 == p.D - 1 && p.D > 1) ? (float)p.z_end : (float)(p.z_start + (double)d1 * p.z_step);
    v3d::world_point(s_ref, xf, yf, z, X, Y, Z);
  }
  const bool live1 = gp1 < P && d1 < p.D;                   // lane 0 is always live
  const bool all_live = __all(live1);
  const float Wm1 = (float)(p.W - 1), Hm1 = (float)(p.H - 1);
  const float rWm1 = p.rWm1, rHm1 = p.rHm1;
  const float Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);
  const int Wp = p.Wf + 2;
  const float Wff = (float)p.Wf, Hff = (float)p.Hf;

  // gather role: 8 lanes x float4 per pixel
  const int gpx = lane >> 3;
  const unsigned cgb = (lane & 7) * 16;
  const unsigned rowb = (unsigned)Wp * CB + cgb;
  const unsigned lane16 = (unsigned)lane * 16u;
  const char* const fb = reinterpret_cast<const char*>(p.featT);
  const char* const wb = reinterpret_cast<const char*>(s_win);
  // LDS byte address of this wave's window (what M0 carries for global_load_lds)
  const unsigned win_lds = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)s_win);
  f32x4 acc_s[kRDB], acc_q[kRDB];
#pragma unroll
  for (int k = 0; k < kRDB; ++k) acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

#if V3D_PSVW_ABLATE == 1
#define V3D_PSV_BLEND(pl_, w_) asm volatile("" : : "v"(t00), "v"(t01), "v"(t10), "v"(t11), "v"(w_) : "memory")
#else
#define V3D_PSV_BLEND(pl_, w_)                                                                       \
  do {                                                                                               \
    f32x4 sv_ = t00 * (w_)[0];                                                                       \
    sv_ = __builtin_elementwise_fma(t01, (f32x4){(w_)[1], (w_)[1], (w_)[1], (w_)[1]}, sv_);          \
    sv_ = __builtin_elementwise_fma(t10, (f32x4){(w_)[2], (w_)[2], (w_)[2], (w_)[2]}, sv_);          \
    /* fourth tap: hipcc copies w[3] into the low half of a register pair first (one v_mov per plane); written by hand  \
       the packed FMA takes the HIGH half of the (w[2], w[3]) pair for both result lanes (op_sel) -- same arithmetic */ \
    {                                                                                                \
      psv_f32x2 lo_ = {sv_[0], sv_[1]}, hi_ = {sv_[2], sv_[3]};                                      \
      const psv_f32x2 w23_ = {(w_)[2], (w_)[3]};                                                     \
      const psv_f32x2 tl_ = {t11[0], t11[1]}, th_ = {t11[2], t11[3]};                                \
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(lo_) : "v"(tl_), "v"(w23_));  \
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(hi_) : "v"(th_), "v"(w23_));  \
      sv_ = (f32x4){lo_[0], lo_[1], hi_[0], hi_[1]};                                                 \
    }                                                                                                \
    acc_s[pl_] += sv_;                                                                               \
    acc_q[pl_] = __builtin_elementwise_fma(sv_, sv_, acc_q[pl_]);                                    \
    /* finish this plane before the next one starts (see the reuse kernel) */                        \
    asm volatile("" : "+v"(acc_s[pl_]), "+v"(acc_q[pl_]) : : "memory");                              \
  } while (0)
#endif

  // camera blocks (P = K [R|t], first cell of the bordered map) of edges e0 .. e0 + 7
  auto load_cams = [&](int e0) __attribute__((always_inline)) {
    const int nload = min(kMaxE, ne - e0) * 12;
#pragma unroll 1
    for (int i = lane; i < nload; i += 64) {
      const int src = p.edge_src[e_begin + e0 + i / 12];
      s_P[i / 12][i % 12] = p.camp[src * kCamStride + 24 + i % 12];
      if (i % 12 == 0) s_base[i / 12] = src * (p.Hf + 2) * Wp;      // cell (-1, -1) of the bordered map
    }
  };
  // projection of this lane's (pixel, plane) sample into the source view of edge e (make_taps, spelled out: the cell
  // coordinates are needed): nw cell of the footprint in the bordered map (x0 + 1, y0 + 1) + the four bilinear weights
  auto project = [&](int e, int& xb, int& yb, f32x4& wq) __attribute__((always_inline)) {
    float ix, iy;
    v3d::sample_position(s_P[e % kMaxE], X, Y, Z, Wm1, rWm1, Hm1, rHm1, Wfm1, Hfm1, ix, iy);
    // clamp to [-1, Wf] x [-1, Hf] (make_taps): v_med3_f32 returns the smallest number when an operand is NaN, i.e. -1,
    // what fminf(fmaxf(NaN, -1), Wf) gives
    ix = __builtin_amdgcn_fmed3f(ix, -1.f, Wff);
    iy = __builtin_amdgcn_fmed3f(iy, -1.f, Hff);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float x1 = x0 + 1.f, y1 = y0 + 1.f;
    const float wx0 = x1 - ix, wx1 = ix - x0, wy0 = y1 - iy, wy1 = iy - y0;
    wq = (f32x4){v3d::mul_rn(wx0, wy0), v3d::mul_rn(wx1, wy0), v3d::mul_rn(wx0, wy1), v3d::mul_rn(wx1, wy1)};
    xb = (int)x0 + 1;
    yb = (int)y0 + 1;
    // lanes beyond the plane grid / the last plane must not stretch the box: they take lane 0's cell (always live) -- read
    // under full exec (inside a branch on !live1 readfirstlane would return the first DEAD lane's own value)
    // (wave-uniform skip: only the waves on the last pixels of the plane grid / the last planes have such lanes)
    if (!all_live) {
      const int xb0 = __builtin_amdgcn_readlane(xb, 0), yb0 = __builtin_amdgcn_readlane(yb, 0);
      xb = live1 ? xb : xb0;
      yb = live1 ? yb : yb0;
    }
  };
  // Software pipeline over the edges: the samples of edge e + 1 are projected while the window of edge e is on its way
  // from L2 to LDS (the copy's latency was 0.27 of 1.59 ms when the wave simply waited for it)
  int xb = 0, yb = 0;
  f32x4 wq = {0.f, 0.f, 0.f, 0.f};
  if (ne > 0) {
    load_cams(0);
    psv_wave_sync<WPB>();
    project(0, xb, yb, wq);
  }
  for (int e = 0; e < ne; ++e) {
    psv_wave_sync<WPB>();                  // the previous pass is done with s_w / s_slot / s_win
#ifdef V3D_PSV_NOPIPE                      // developer A/B: project at the top of the pass, wait for the copy right after issuing it
    if (e > 0) {
      if (e % kMaxE == 0) { load_cams(e); psv_wave_sync<WPB>(); }
      project(e, xb, yb, wq);
    }
#endif
    // Window = the box spanned by the four corner samples (first / last pixel on the first / last plane: a projective map is
    // monotone in pixel and in depth, so they bound the 64 footprints in all but a few tiles), cut to 16 x 4 cells.  A
    // sample whose footprint is not inside it is simply marked (bit 0 of its tap word) and takes its cells from featT.
    int xmin, ymin, ncol, nrow;            // wave-uniform
    {
      // (both coordinates of a corner travel in one word: four lane reads instead of eight; bordered coordinates are >= 0)
      const int cc = (yb << 16) | xb;
      const int ca = __builtin_amdgcn_readlane(cc, 0), cb = __builtin_amdgcn_readlane(cc, 7);
      const int cd = __builtin_amdgcn_readlane(cc, 56), ce = __builtin_amdgcn_readlane(cc, 63);
      const int xa = ca & 0xffff, xc = cb & 0xffff, xd = cd & 0xffff, xe = ce & 0xffff;
      const int ya = ca >> 16, yc = cb >> 16, yd = cd >> 16, ye = ce >> 16;
      xmin = psv_smin(psv_smin(xa, xc), psv_smin(xd, xe));
      ymin = psv_smin(psv_smin(ya, yc), psv_smin(yd, ye));
      ncol = psv_smax(psv_smax(xa, xc), psv_smax(xd, xe)) - xmin + 2 > 8 ? kWinCols : 8;
      nrow = psv_smin(psv_smax(psv_smax(ya, yc), psv_smax(yd, ye)) - ymin + 2, kWinRows);
    }
    const int base_cell = __builtin_amdgcn_readfirstlane(s_base[e % kMaxE]);
    const unsigned dx = (unsigned)(xb - xmin), dy = (unsigned)(yb - ymin);
    const bool inwin = dx <= (unsigned)(ncol - 2) && dy <= (unsigned)(nrow - 2);
    // tap word: byte offset of the nw cell in the window, or -- sign bit set -- in featT (< 2 GB, checked by the host)
    const unsigned so_win = ((dy << 4) | dx) * CB;
    const unsigned so_ext = ((unsigned)(base_cell + __mul24(yb, Wp) + xb) * CB) | 0x80000000u;
    const unsigned so = inwin ? so_win : so_ext;
    s_w[lane] = wq;
    s_slot[lane] = so;
    // does ANY pixel change its footprint on plane k (bit group k of the ballot)?  plane 0 always loads
    const unsigned prev = (unsigned)__builtin_amdgcn_ds_bpermute(((lane - 8) & 63) * 4, (int)so);
    const unsigned long long chg = __ballot(lane < 8 || so != prev);
    // copy the window: nrow rows of 8 or 16 cells from (xmin, ymin) (runs past the last needed column stay inside the bordered
    // maps + tail); lane l moves bytes [16 l, 16 l + 16) of each 1 KB run
    {
      // wave-uniform row address (SGPR pair) + the lane's 16 bytes (one constant VGPR): no per-lane address arithmetic.  The
      // copies are written in assembly (hipcc forms 64-bit per-lane addresses for the builtin): M0 = LDS byte address of the
      // run, saved and restored around each copy; the wave waits for them itself (vmcnt below).
      const char* rowp = fb + (size_t)((unsigned)(base_cell + ymin * Wp + xmin) * CB);
      unsigned dst = win_lds;
      asm volatile("s_nop 4" ::: "memory");            // SGPRs written by v_readlane may feed the first copy's address
#pragma unroll 1
      for (int rr = 0; rr < (V3D_PSVW_ABLATE == 3 ? 0 : nrow); ++rr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(rowp), "s"(dst) : "memory");
        if (ncol > 8)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(lane16), "s"(rowp), "s"(dst) : "memory");   // the instruction offset moves BOTH addresses
        rowp += (size_t)Wp * CB;
        dst += kWinCols * CB;
      }
    }
#ifndef V3D_PSV_NOPIPE
    if (e + 1 < ne) {
      if ((e + 1) % kMaxE == 0) {
        psv_wave_sync<WPB>();
        load_cams(e + 1);
        psv_wave_sync<WPB>();
      }
      project(e + 1, xb, yb, wq);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the window has landed, the tap records are written
    psv_wave_sync<WPB>();
    // deliberately not initialised: the first plane of a pass always loads them
    f32x4 t00, t01, t10, t11;
    f32x4 wn = s_w[gpx];
    unsigned sn = s_slot[gpx];
#pragma unroll
    for (int pl = 0; pl < kRDB; ++pl) {
      const f32x4 w = wn;
      const unsigned so_pl = sn;
      if (pl + 1 < kRDB) {               // next plane's record: in flight during this plane's blend
        wn = s_w[(pl + 1) * kRPix + gpx];
        sn = s_slot[(pl + 1) * kRPix + gpx];
      }
      if (((chg >> (8 * pl)) & 0xffull) && (V3D_PSVW_ABLATE != 2 || (pl == 0 && e == 0))) {      // wave-uniform: all pixels reload together (a load costs the same masked or not)
        // (a wave-uniform test "no sample of this plane is outside the window" -- one ballot per pass -- in front of the per-lane
        // sign test saves a v_cmp and the exec-mask juggling in 95 % of the planes and measured 0.6 % SLOWER, round 4)
        if (V3D_PSVW_ABLATE != 6 && (int)so_pl < 0) {
          const unsigned b00 = so_pl & 0x7fffffffu;
          t00 = *reinterpret_cast<const f32x4*>(fb + (size_t)(b00 + cgb));
          t01 = *reinterpret_cast<const f32x4*>(fb + (size_t)(b00 + cgb) + CB);
          t10 = *reinterpret_cast<const f32x4*>(fb + (size_t)(b00 + rowb));
          t11 = *reinterpret_cast<const f32x4*>(fb + (size_t)(b00 + rowb) + CB);
        } else {
          const char* a = wb + (so_pl + cgb);
          t00 = *reinterpret_cast<const f32x4*>(a);
          t01 = *reinterpret_cast<const f32x4*>(a + CB);
          t10 = *reinterpret_cast<const f32x4*>(a + kWinCols * CB);
          t11 = *reinterpret_cast<const f32x4*>(a + kWinCols * CB + CB);
        }
      }
      V3D_PSV_BLEND(pl, w);
    }
  }
#undef V3D_PSV_BLEND

  // ---- variance -> stores (as in the reuse kernel) -------------------------------------------------------------
  psv_wave_sync<WPB>();
  const float cnt = (float)max(ne, 1);        // torch_scatter mean: sum / clamp(count, 1)
  const bool cnt_pow2 = (max(ne, 1) & (max(ne, 1) - 1)) == 0;
  const float cnt_inv = 1.f / cnt;
  const int cg = lane & 7;
  if constexpr (SPLIT) {
    const int chunk = cg >> 1, half = cg & 1;
    u32x4* const out = reinterpret_cast<u32x4*>(p.var);
    const int gp = ptile * kRPix + gpx;
    // the lane's slot on the chunk's first plane; plane pl is pl * P slots further (a wave-uniform stride: with the whole
    // index spelled out per plane the compiler rebuilt the 64-bit product -- two v_mul_lo_u32 and a v_mad_u64_u32 -- eight times)
    u32x4* const obase = out + (((size_t)r * 8 + chunk * 2 + half) * p.D + (size_t)dchunk * kRDB) * P + gp;
    // x / cnt is an IEEE division; for a power-of-two count x * (1 / cnt) is the same number exactly.  ONE wave-uniform
    // branch around the whole epilogue (the reuse kernel tests it per value: 64 branches)
#define V3D_PSV_EMIT(MEAN)                                                                                              \
  _Pragma("unroll") for (int pl = 0; pl < kRDB; ++pl) {                                                                 \
    float v[4];                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                     \
      const float avg = MEAN(acc_s[pl][k]);                                                                             \
      const float avg_sq = MEAN(acc_q[pl][k]);                                                                          \
      v[k] = v3d::sub_rn(avg_sq, v3d::mul_rn(avg, avg));                    /* mvsnet.py:216 */                         \
    }                                                                                                                   \
    const int d = dchunk * kRDB + pl;                                                                                   \
    if constexpr (CL8) {                                                                                                \
      /* channels 4 cg .. 4 cg + 3 = half `half` of the voxel's channel group `chunk`: the lane's own 16 bytes */         \
      const u32x4 slot = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};  \
      if (gp < P && d < p.D)                                                                                            \
        __builtin_nontemporal_store(slot, obase + (size_t)pl * (unsigned)P);                                             \
    } else {                                                                                                            \
    const unsigned h01 = pack_bf16x2(v[0], v[1]), h23 = pack_bf16x2(v[2], v[3]);                                        \
    const unsigned l01 = pack_bf16x2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));     \
    const unsigned l23 = pack_bf16x2(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));     \
    const unsigned s0 = half ? h01 : l01, s1 = half ? h23 : l23;             /* what the partner lane stores */          \
    const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xf, 0xf, true);   /* quad_perm [1,0,3,2] */   \
    const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xf, 0xf, true);                              \
    const u32x4 slot = half ? (u32x4){r0, r1, l01, l23} : (u32x4){h01, h23, r0, r1};                                    \
    if (gp < P && d < p.D && (V3D_PSVW_ABLATE != 5 || slot[0] == 0x12345u))                                             \
      __builtin_nontemporal_store(slot, obase + (size_t)pl * (unsigned)P);                                               \
    }                                                                                                                   \
  }
#define V3D_MEAN_MUL(x) ((x) * cnt_inv)
    // any other count: the correctly rounded quotient from the correctly rounded reciprocal (v3d::div_uniform, Markstein;
    // 4 instructions instead of the 11 of the IEEE sequence, twice per output value); v_div_fixup restores the special cases
#define V3D_MEAN_DIV(x) __builtin_amdgcn_div_fixupf(v3d::div_uniform((x), cnt, cnt_inv), cnt, (x))
    if (cnt_pow2) { V3D_PSV_EMIT(V3D_MEAN_MUL) } else { V3D_PSV_EMIT(V3D_MEAN_DIV) }
#undef V3D_PSV_EMIT
#undef V3D_MEAN_MUL
#undef V3D_MEAN_DIV
  } else {
    constexpr int NPX = WPB * kRPix;
    float (*const s_out)[C][NPX + 1] = reinterpret_cast<float (*)[C][NPX + 1]>(&s_win_[0][0]);
    auto mean = [&](float x) __attribute__((always_inline)) {
      return cnt_pow2 ? x * cnt_inv : __builtin_amdgcn_div_fixupf(v3d::div_uniform(x, cnt, cnt_inv), cnt, x);
    };
    const int gp0 = (ptile - wv) * kRPix;             // first pixel of the workgroup
#pragma unroll
    for (int round = 0; round < kRDB / kOutPl; ++round) {
      __syncthreads();                                 // every wave is done with its window / the previous round is stored
#pragma unroll
      for (int q = 0; q < kOutPl; ++q) {
        const int pl = round * kOutPl + q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float avg = mean(acc_s[pl][k]);
          const float avg_sq = mean(acc_q[pl][k]);
          s_out[q][cg * 4 + k][wv * kRPix + gpx] = v3d::sub_rn(avg_sq, v3d::mul_rn(avg, avg));   // mvsnet.py:216
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < kOutPl * C * NPX; i += 64 * WPB) {
        const int q = i / (C * NPX), c = (i / NPX) % C, px = i % NPX;
        const int gp = gp0 + px, d = dchunk * kRDB + round * kOutPl + q;
        if (gp < P && d < p.D) __builtin_nontemporal_store(s_out[q][c][px], &p.var[(((size_t)r * C + c) * p.D + d) * P + gp]);
      }
    }
  }
}

// Diagnostic twin of the warp kernels' projection (v3d_psv_sample_positions_f32): one thread per (edge, plane, pixel)
// runs the SAME device functions (v3d::world_point / v3d::sample_position) and stores the un-normalised sample position,
// so a test can compare the coordinates the kernels use with the reference's, bit for bit.
__global__ __launch_bounds__(256) void psv_positions_kernel(PsvParams p, float* __restrict__ pos, float* __restrict__ world) {
  const int P = p.h * p.w;
  const long long n_vox = (long long)p.D * P;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (i >= n_vox) return;
  const int d = (int)(i / P), gp = (int)(i % P), gy = gp / p.w, gx = gp % p.w;
  const float xf = (p.w > 1 && gx == p.w - 1) ? (float)(p.W - 1) : (float)((double)gx * p.x_step);
  const float yf = (p.h > 1 && gy == p.h - 1) ? (float)(p.H - 1) : (float)((double)gy * p.y_step);
  const float z = (d == p.D - 1 && p.D > 1) ? (float)p.z_end : (float)(p.z_start + (double)d * p.z_step);
  float X, Y, Z;
  v3d::world_point(p.camp + p.ref_img[r] * kCamStride, xf, yf, z, X, Y, Z);
  if (world) {
    world[((size_t)r * 3 + 0) * n_vox + i] = X;
    world[((size_t)r * 3 + 1) * n_vox + i] = Y;
    world[((size_t)r * 3 + 2) * n_vox + i] = Z;
  }
  const float Wm1 = (float)(p.W - 1), Hm1 = (float)(p.H - 1);
  const float rWm1 = (float)(1.0 / (double)(p.W - 1)), rHm1 = (float)(1.0 / (double)(p.H - 1));
  const float Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);
  for (int e = p.edge_ofs[r]; e < p.edge_ofs[r + 1]; ++e) {
    float ix, iy;
    v3d::sample_position(p.camp + p.edge_src[e] * kCamStride + 24, X, Y, Z, Wm1, rWm1, Hm1, rHm1, Wfm1, Hfm1, ix, iy);
    pos[((size_t)e * n_vox + i) * 2] = ix;
    pos[((size_t)e * n_vox + i) * 2 + 1] = iy;
  }
}

}  // namespace

// shared with backproject.hip
int v3d::transpose_channel_last(const float* feat, float* featT, int n_img, int C, int HW, hipStream_t s) {
  dim3 tg((HW + kPix - 1) / kPix, n_img);
  v3d::TimedScope ts("transpose_channel_last", s);
  if (C == 32) transpose_channel_last_kernel<32><<<tg, 256, 0, s>>>(feat, featT, HW);
  else if (C == 16) transpose_channel_last_kernel<16><<<tg, 256, 0, s>>>(feat, featT, HW);
  else return -1;
  return 0;
}

// bordered channel-last feature maps: n_img x (Hf + 2) x (Wf + 2) cells + the tail of zero cells behind them
static size_t psv_feat_bytes(int n_img, int C, int Hf, int Wf) {
  return ((size_t)n_img * (Hf + 2) * (Wf + 2) + kTailCells(Wf)) * C * sizeof(float);
}

extern "C" size_t v3d_psv_workspace_bytes(int n_img, int C, int Hf, int Wf) {
  // channel-last copy of the feature maps + the per-image camera blocks
  return v3d::align_up(psv_feat_bytes(n_img, C, Hf, Wf), 256) + v3d::align_up((size_t)n_img * kCamStride * sizeof(float), 256);
}

static int psv_variance_impl(int mode, const float* feat, const float* K, const float* R,
                                    const float* t, const int32_t* ref_img,
                                    const int32_t* edge_ofs, const int32_t* edge_src, int n_img,
                                    int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                                    double depth_start, double depth_interval, int D, int h, int w,
                                    float* var, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  const bool split = mode != 0, cl8 = mode == 2;      // mode: 0 reference layout, 1 split-bf16 hand-off, 2 fp32 channel-last
  V3D_REQUIRE(feat && K && R && t && ref_img && edge_ofs && edge_src && var && workspace,
              V3D_ERR_BAD_ARG, "v3d_psv_variance_f32: null pointer argument");
  V3D_REQUIRE(C == 32 || C == 16, V3D_ERR_UNSUPPORTED,
              "v3d_psv_variance_f32: C=%d unsupported (16 or 32)", C);
  // the reuse kernel divides by W - 1 and H - 1 through their correctly rounded reciprocals (exact unless the divisor's
  // significand is all ones, i.e. >= 2^24 - 1)
  V3D_REQUIRE(W <= (1 << 20) && H <= (1 << 20), V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: image size out of range");
  V3D_REQUIRE(!split || C == 32, V3D_ERR_UNSUPPORTED,
              "v3d_psv_variance_split: C=%d unsupported (the split layout is defined for 32 channels)", C);
  V3D_REQUIRE(n_img > 0 && n_ref > 0 && n_edges >= 0 && Hf > 0 && Wf > 0 && H > 1 && W > 1 &&
                  D > 0 && h > 0 && w > 0,
              V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: bad shape");
  V3D_REQUIRE(Hf < 32760 && Wf < 32760, V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: feature maps larger than 32759 cells per axis");
  V3D_REQUIRE(psv_feat_bytes(n_img, C, Hf, Wf) < (size_t)1 << 32, V3D_ERR_BAD_SHAPE,
              "v3d_psv_variance_f32: bordered feature maps exceed 4 GB (32-bit byte offsets)");
  V3D_REQUIRE(workspace_bytes >= v3d_psv_workspace_bytes(n_img, C, Hf, Wf),
              V3D_ERR_WORKSPACE_TOO_SMALL, "v3d_psv_variance_f32: workspace %zu < %zu",
              workspace_bytes, v3d_psv_workspace_bytes(n_img, C, Hf, Wf));
  hipStream_t s = (hipStream_t)stream;
  float* featT = (float*)workspace;
  float* camp = (float*)((char*)workspace + v3d::align_up(psv_feat_bytes(n_img, C, Hf, Wf), 256));
  {
    v3d::TimedScope ts("transpose_channel_last", s);
    const dim3 tg(((Hf + 2) * (Wf + 2) + kPix - 1) / kPix, n_img);
    if (C == 32) transpose_bordered_kernel<32><<<tg, 256, 0, s>>>(feat, featT, Hf, Wf, n_img);
    else transpose_bordered_kernel<16><<<tg, 256, 0, s>>>(feat, featT, Hf, Wf, n_img);
  }
  V3D_CHECK_LAUNCH("transpose_bordered_kernel");
  cam_setup_kernel<<<(n_img + 63) / 64, 64, 0, s>>>(K, R, t, camp, n_img);
  V3D_CHECK_LAUNCH("cam_setup_kernel");

  PsvParams p;
  p.featT = featT; p.K = K; p.R = R; p.t = t;
  p.ref_img = ref_img; p.edge_ofs = edge_ofs; p.edge_src = edge_src; p.var = var; p.camp = camp;
  p.n_img = n_img; p.n_ref = n_ref; p.Hf = Hf; p.Wf = Wf; p.H = H; p.W = W; p.D = D;
  p.h = h; p.w = w;
  // workgroup size: single-wave workgroups (16 pixels) unless the developer option psv_threads = 256 asks for the 64-pixel variant
  const int threads = v3d::option(v3d::kOptPsvThreads);
  V3D_REQUIRE(threads == 64 || threads == 256, V3D_ERR_BAD_ARG, "option psv_threads must be 64 or 256");
  const int pix = threads / 4;
  p.n_ptile = (h * w + pix - 1) / pix;
  p.x_step = w > 1 ? (double)(W - 1) / (double)(w - 1) : 0.0;
  p.y_step = h > 1 ? (double)(H - 1) / (double)(h - 1) : 0.0;
  const double depth_end = depth_start + (double)(D - 1) * depth_interval;
  p.z_start = depth_start;
  p.z_step = D > 1 ? (depth_end - depth_start) / (double)(D - 1) : 0.0;
  p.z_end = depth_end;
  const int n_dchunk = (D + kDB - 1) / kDB;
  const long long blocks = (long long)n_ref * n_dchunk * p.n_ptile;
  V3D_REQUIRE(blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: grid too large");
  {
    v3d::TimedScope ts("psv_variance", s);
    const unsigned grid = (unsigned)blocks;
#define V3D_PSV(C_, SPLIT_)                                                        \
  do {                                                                             \
    if (threads == 64) psv_variance_kernel<C_, SPLIT_, 64><<<grid, 64, 0, s>>>(p); \
    else psv_variance_kernel<C_, SPLIT_, 256><<<grid, 256, 0, s>>>(p);             \
  } while (0)
    const int psv_kernel = v3d::option(v3d::kOptPsvKernel);     // developer A/B (v3d_set_option): 0 auto, 1 reuse, 2 gather kernel
    const bool plain_gather = psv_kernel == 2;
    if (C == 32 && !plain_gather) {
      const long long rblocks = (long long)n_ref * ((D + kRDB - 1) / kRDB) * ((h * w + kRPix - 1) / kRPix);
      V3D_REQUIRE(rblocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: grid too large");
      p.n_ptile = (h * w + kRPix - 1) / kRPix;
      const bool reuse_env = psv_kernel == 1 || kRDB != 8;   // (the round-2 kernel)
      // the window kernel's tap words keep their sign bit as a flag: feature maps beyond 2 GB take the reuse kernel
      const bool no_window = reuse_env || psv_feat_bytes(n_img, C, Hf, Wf) >= ((size_t)1 << 31);
      V3D_REQUIRE(!cl8 || !no_window, V3D_ERR_UNSUPPORTED,
                  "v3d_psv_variance_cl8: only the window kernel writes this layout (feature maps < 2 GB, no developer switch)");
      {      // launch constants of the window kernel (PsvParams)
        const unsigned ndc = (unsigned)((D + kRDB - 1) / kRDB);
        const unsigned npt = (unsigned)(split || cl8 ? p.n_ptile : (p.n_ptile + 3) / 4);
        const unsigned long long nblk = (unsigned long long)n_ref * ndc * npt;
        p.rWm1 = (float)(1.0 / (double)(W - 1));
        p.rHm1 = (float)(1.0 / (double)(H - 1));
        p.m_dchunk = v3d::magic_u32(nblk, ndc);
        p.m_ptile = v3d::magic_u32(nblk / ndc + 1, npt);
        p.m_w = v3d::magic_u32((unsigned long long)h * w + 4 * kRPix, (unsigned)w);
      }
      if (cl8) {
        psv_variance_window_kernel<true, true><<<(unsigned)rblocks, 64, 0, s>>>(p);
      } else if (split) {
        if (no_window) psv_variance_reuse_kernel<true><<<(unsigned)rblocks, 64, 0, s>>>(p);
        else psv_variance_window_kernel<true><<<(unsigned)rblocks, 64, 0, s>>>(p);
      } else {      // four 8-pixel tiles per workgroup
        p.n_ptile = (p.n_ptile + 3) / 4;
        const long long fblocks = (long long)n_ref * ((D + kRDB - 1) / kRDB) * p.n_ptile;
        if (no_window) psv_variance_reuse_kernel<false><<<(unsigned)fblocks, 256, 0, s>>>(p);
        else psv_variance_window_kernel<false><<<(unsigned)fblocks, 256, 0, s>>>(p);
      }
    } else if (cl8) {
      return v3d::fail(V3D_ERR_UNSUPPORTED, "v3d_psv_variance_cl8: C=%d unsupported (32) / option psv_kernel = 2", C);
    } else if (split) V3D_PSV(32, true);
    else if (C == 32) V3D_PSV(32, false);
    else V3D_PSV(16, false);
#undef V3D_PSV
  }
  V3D_CHECK_LAUNCH("psv_variance_kernel");
  return V3D_OK;
}

extern "C" int v3d_psv_sample_positions_f32(const float* K, const float* R, const float* t, const int32_t* ref_img,
                                            const int32_t* edge_ofs, const int32_t* edge_src, int n_img, int n_ref,
                                            int n_edges, int Hf, int Wf, int H, int W, double depth_start,
                                            double depth_interval, int D, int h, int w, float* pos, float* world,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  V3D_REQUIRE(K && R && t && ref_img && edge_ofs && edge_src && pos && workspace, V3D_ERR_BAD_ARG,
              "v3d_psv_sample_positions_f32: null pointer argument");
  V3D_REQUIRE(n_img > 0 && n_ref > 0 && n_edges >= 0 && Hf > 0 && Wf > 0 && H > 1 && W > 1 && D > 0 && h > 0 && w > 0,
              V3D_ERR_BAD_SHAPE, "v3d_psv_sample_positions_f32: bad shape");
  V3D_REQUIRE(workspace_bytes >= (size_t)n_img * kCamStride * sizeof(float), V3D_ERR_WORKSPACE_TOO_SMALL,
              "v3d_psv_sample_positions_f32: workspace too small (n_img * 36 floats)");
  hipStream_t s = (hipStream_t)stream;
  float* camp = (float*)workspace;
  cam_setup_kernel<<<(n_img + 63) / 64, 64, 0, s>>>(K, R, t, camp, n_img);
  PsvParams p;
  memset(&p, 0, sizeof(p));
  p.ref_img = ref_img; p.edge_ofs = edge_ofs; p.edge_src = edge_src; p.camp = camp;
  p.n_img = n_img; p.n_ref = n_ref; p.Hf = Hf; p.Wf = Wf; p.H = H; p.W = W; p.D = D; p.h = h; p.w = w;
  p.x_step = w > 1 ? (double)(W - 1) / (double)(w - 1) : 0.0;
  p.y_step = h > 1 ? (double)(H - 1) / (double)(h - 1) : 0.0;
  const double depth_end = depth_start + (double)(D - 1) * depth_interval;
  p.z_start = depth_start;
  p.z_step = D > 1 ? (depth_end - depth_start) / (double)(D - 1) : 0.0;
  p.z_end = depth_end;
  const long long n_vox = (long long)D * h * w;
  psv_positions_kernel<<<dim3((unsigned)((n_vox + 255) / 256), n_ref), 256, 0, s>>>(p, pos, world);
  V3D_CHECK_LAUNCH("psv_positions_kernel");
  return V3D_OK;
}

extern "C" int v3d_psv_variance_f32(const float* feat, const float* K, const float* R, const float* t,
                                    const int32_t* ref_img, const int32_t* edge_ofs, const int32_t* edge_src,
                                    int n_img, int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                                    double depth_start, double depth_interval, int D, int h, int w, float* var,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return psv_variance_impl(0, feat, K, R, t, ref_img, edge_ofs, edge_src, n_img, n_ref, n_edges, C, Hf, Wf, H, W,
                           depth_start, depth_interval, D, h, w, var, workspace, workspace_bytes, stream);
}

extern "C" int v3d_psv_variance_split(const float* feat, const float* K, const float* R, const float* t,
                                      const int32_t* ref_img, const int32_t* edge_ofs, const int32_t* edge_src,
                                      int n_img, int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                                      double depth_start, double depth_interval, int D, int h, int w,
                                      void* var_split, void* workspace, size_t workspace_bytes, void* stream) {
  return psv_variance_impl(1, feat, K, R, t, ref_img, edge_ofs, edge_src, n_img, n_ref, n_edges, C, Hf, Wf, H, W,
                           depth_start, depth_interval, D, h, w, (float*)var_split, workspace, workspace_bytes,
                           stream);
}

extern "C" int v3d_psv_variance_cl8(const float* feat, const float* K, const float* R, const float* t,
                                    const int32_t* ref_img, const int32_t* edge_ofs, const int32_t* edge_src,
                                    int n_img, int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                                    double depth_start, double depth_interval, int D, int h, int w,
                                    float* var_cl8, void* workspace, size_t workspace_bytes, void* stream) {
  return psv_variance_impl(2, feat, K, R, t, ref_img, edge_ofs, edge_src, n_img, n_ref, n_edges, C, Hf, Wf, H, W,
                           depth_start, depth_interval, D, h, w, var_cl8, workspace, workspace_bytes, stream);
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_psv_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)8 * kPhaseSlots);
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_psv_phase), h.size() * sizeof(unsigned long long)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < kPhaseSlots; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif
