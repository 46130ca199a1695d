// Rows A1-A4 of SURVEY.md §8a: plane-sweep homography warp of source-view features + cross-view
// variance, fused (no world points, sampling grid, warped volume or squared volume is ever
// materialised).  Reference semantics: mv3d/utils.py:86-108, mv3d/subnetworks/mvsnet.py:192-216.
//
// Data layout in HBM
//   feat   [n_img, C, Hf, Wf]  (reference layout)  -> transposed once per call into
//   featT  [n_img, Hf, Wf, C]  (workspace) so that one bilinear tap of all C channels is one
//                               contiguous 4*C-byte run (C=32: exactly one 128-B cache line);
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
  int n_img, n_ref, Hf, Wf, H, W, D, h, w, n_ptile;
  double x_step, y_step, z_start, z_step, z_end;
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

struct TapInfo {      // one (pixel, edge) pair, 32 bytes
  float w00, w01, w10, w11;   // nw, ne, sw, se weights (0 where the tap is out of range)
  int o00, o01, o10, o11;     // element offsets of the 4 taps into featT (already * C); -1 = skip all
};

template <int C>
__global__ __launch_bounds__(kThreads) void psv_variance_kernel(PsvParams p) {
  constexpr int LP = C / 4;               // lanes per pixel
  constexpr int PPP = kThreads / LP;      // pixels per phase-2 pass
  constexpr int NPASS = kPix / PPP;
  static_assert(kPix % PPP == 0, "tile");

  __shared__ TapInfo s_tap[kMaxE * kPix];
  __shared__ float s_out[C][kPix + 1];
  __shared__ float s_ref[24];             // Kinv(9) R(9) t(3)
  __shared__ float s_P[kMaxE][12];

  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int ptile = b % p.n_ptile; b /= p.n_ptile;
  const int n_dchunk = (p.D + kDB - 1) / kDB;
  const int dchunk = b % n_dchunk;
  const int r = b / n_dchunk;
  const int P = p.h * p.w;
  const int e_begin = p.edge_ofs[r], e_end = p.edge_ofs[r + 1];
  const int ne = e_end - e_begin;
  const int ref = p.ref_img[r];

  if (tid == 0) {
    // K^-1 in fp64 (adjugate / determinant), rounded to f32 (utils.py:103 torch.inverse)
    const float* Kp = p.K + ref * 9;
    double a = Kp[0], bb = Kp[1], c = Kp[2], d = Kp[3], e = Kp[4], f = Kp[5], g = Kp[6],
           hh = Kp[7], i = Kp[8];
    double det = a * (e * i - f * hh) - bb * (d * i - f * g) + c * (d * hh - e * g);
    double id = 1.0 / det;
    s_ref[0] = (float)((e * i - f * hh) * id);
    s_ref[1] = (float)((c * hh - bb * i) * id);
    s_ref[2] = (float)((bb * f - c * e) * id);
    s_ref[3] = (float)((f * g - d * i) * id);
    s_ref[4] = (float)((a * i - c * g) * id);
    s_ref[5] = (float)((c * d - a * f) * id);
    s_ref[6] = (float)((d * hh - e * g) * id);
    s_ref[7] = (float)((bb * g - a * hh) * id);
    s_ref[8] = (float)((a * e - bb * d) * id);
  }
  if (tid >= 64 && tid < 64 + 9) s_ref[9 + tid - 64] = p.R[ref * 9 + tid - 64];
  if (tid >= 128 && tid < 128 + 3) s_ref[18 + tid - 128] = p.t[ref * 3 + tid - 128];

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
  const float Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);

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
      // projection matrices P = K [R|t] of this chunk's source views (mvsnet.py:196-197)
      if (tid < nec * 12) {
        int e = tid / 12, ij = tid % 12, i = ij / 4, j = ij % 4;
        int src = p.edge_src[e_begin + ec + e];
        const float* Kp = p.K + src * 9;
        const float* Rp = p.R + src * 9;
        const float* tp = p.t + src * 3;
        float v;
        if (j < 3) v = Kp[i * 3 + 0] * Rp[0 * 3 + j] + Kp[i * 3 + 1] * Rp[1 * 3 + j] + Kp[i * 3 + 2] * Rp[2 * 3 + j];
        else v = Kp[i * 3 + 0] * tp[0] + Kp[i * 3 + 1] * tp[1] + Kp[i * 3 + 2] * tp[2];
        s_P[e][ij] = v;
      }
      __syncthreads();
      // ---- phase 1: project (pixel, edge) pairs ------------------------------------------
      {
        // world point of (pixel, plane): X = R^T (K^-1 [x z, y z, z] - t)   (utils.py:98-106)
        float p0 = xf * z, p1 = yf * z, p2 = z;
        float c0 = s_ref[0] * p0 + s_ref[1] * p1 + s_ref[2] * p2 - s_ref[18];
        float c1 = s_ref[3] * p0 + s_ref[4] * p1 + s_ref[5] * p2 - s_ref[19];
        float c2 = s_ref[6] * p0 + s_ref[7] * p1 + s_ref[8] * p2 - s_ref[20];
        float X = s_ref[9] * c0 + s_ref[12] * c1 + s_ref[15] * c2;
        float Y = s_ref[10] * c0 + s_ref[13] * c1 + s_ref[16] * c2;
        float Z = s_ref[11] * c0 + s_ref[14] * c1 + s_ref[17] * c2;
        for (int e = tid / kPix; e < nec; e += kThreads / kPix) {
          const float* Pm = s_P[e];
          float qx = Pm[0] * X + Pm[1] * Y + Pm[2] * Z + Pm[3];
          float qy = Pm[4] * X + Pm[5] * Y + Pm[6] * Z + Pm[7];
          float qz = Pm[8] * X + Pm[9] * Y + Pm[10] * Z + Pm[11];
          float zb = fabsf(qz) + 1e-8f;                        // mvsnet.py:200-201
          float u = qx / zb, v = qy / zb;
          float gx = (u / Wm1) * 2.f - 1.f;                    // mvsnet.py:205-206
          float gy = (v / Hm1) * 2.f - 1.f;
          float ix = ((gx + 1.f) / 2.f) * Wfm1;                // grid_sample, align_corners=True
          float iy = ((gy + 1.f) / 2.f) * Hfm1;
          float x0 = floorf(ix), y0 = floorf(iy);
          float x1 = x0 + 1.f, y1 = y0 + 1.f;
          bool vx0 = (x0 >= 0.f) && (x0 <= Wfm1), vx1 = (x1 >= 0.f) && (x1 <= Wfm1);
          bool vy0 = (y0 >= 0.f) && (y0 <= Hfm1), vy1 = (y1 >= 0.f) && (y1 <= Hfm1);
          TapInfo ti;
          ti.w00 = (vx0 && vy0) ? (x1 - ix) * (y1 - iy) : 0.f;
          ti.w01 = (vx1 && vy0) ? (ix - x0) * (y1 - iy) : 0.f;
          ti.w10 = (vx0 && vy1) ? (x1 - ix) * (iy - y0) : 0.f;
          ti.w11 = (vx1 && vy1) ? (ix - x0) * (iy - y0) : 0.f;
          bool any = (vx0 || vx1) && (vy0 || vy1) && (gp1 < P);
          int xi0 = vx0 ? (int)x0 : 0, xi1 = vx1 ? (int)x1 : 0;
          int yi0 = vy0 ? (int)y0 : 0, yi1 = vy1 ? (int)y1 : 0;
          int base = p.edge_src[e_begin + ec + e] * p.Hf * p.Wf;
          ti.o00 = any ? (base + yi0 * p.Wf + xi0) * C : -1;
          ti.o01 = (base + yi0 * p.Wf + xi1) * C;
          ti.o10 = (base + yi1 * p.Wf + xi0) * C;
          ti.o11 = (base + yi1 * p.Wf + xi1) * C;
          s_tap[e * kPix + px1] = ti;
        }
      }
      __syncthreads();
      // ---- phase 2: gather + accumulate ----------------------------------------------------
#pragma unroll
      for (int a = 0; a < NPASS; ++a) {
        const int px = a * PPP + pix_in_pass;
        for (int e = 0; e < nec; ++e) {
          const TapInfo ti = s_tap[e * kPix + px];
          float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ti.o00 >= 0) {
            const float4 v00 = *reinterpret_cast<const float4*>(p.featT + ti.o00 + cg * 4);
            const float4 v01 = *reinterpret_cast<const float4*>(p.featT + ti.o01 + cg * 4);
            const float4 v10 = *reinterpret_cast<const float4*>(p.featT + ti.o10 + cg * 4);
            const float4 v11 = *reinterpret_cast<const float4*>(p.featT + ti.o11 + cg * 4);
            s.x = v00.x * ti.w00; s.y = v00.y * ti.w00; s.z = v00.z * ti.w00; s.w = v00.w * ti.w00;
            s.x += v01.x * ti.w01; s.y += v01.y * ti.w01; s.z += v01.z * ti.w01; s.w += v01.w * ti.w01;
            s.x += v10.x * ti.w10; s.y += v10.y * ti.w10; s.z += v10.z * ti.w10; s.w += v10.w * ti.w10;
            s.x += v11.x * ti.w11; s.y += v11.y * ti.w11; s.z += v11.z * ti.w11; s.w += v11.w * ti.w11;
          }
          acc_s[a][0] += s.x; acc_s[a][1] += s.y; acc_s[a][2] += s.z; acc_s[a][3] += s.w;
          acc_q[a][0] += s.x * s.x; acc_q[a][1] += s.y * s.y;
          acc_q[a][2] += s.z * s.z; acc_q[a][3] += s.w * s.w;
        }
      }
    }
    // ---- variance, transpose through LDS, coalesced store -------------------------------------
    const float cnt = (float)max(ne, 1);        // torch_scatter mean: sum / clamp(count, 1)
#pragma unroll
    for (int a = 0; a < NPASS; ++a) {
      const int px = a * PPP + pix_in_pass;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float avg = acc_s[a][k] / cnt;
        float avg_sq = acc_q[a][k] / cnt;
        s_out[cg * 4 + k][px] = __fsub_rn(avg_sq, __fmul_rn(avg, avg));   // mvsnet.py:216
      }
    }
    __syncthreads();
    {
      const int px = tid % kPix;
      const int gp = ptile * kPix + px;
      if (gp < P) {
        for (int c = tid / kPix; c < C; c += kThreads / kPix)
          p.var[(((size_t)r * C + c) * p.D + d) * P + gp] = s_out[c][px];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// LDS-window variant (C == 32): one 512-thread workgroup per (reference view, 8x8 plane-grid pixels,
// 8 depth planes).  Thread t owns sample (plane t/64, pixel t%64) in the projection phase and
// (pixel t/8, channel group t%8) in the sampling phase.  Per source edge:
//   A  project the 512 samples, store (ix, iy) in LDS, reduce their cell bounding box;
//   B  if the window (<= kWinCells feature cells) fits, copy it featT -> LDS with coalesced float4 loads
//      (each 128-B cell is read ONCE for all the taps that touch it: ~8x less L1 traffic than gathering
//      4 x 128 B per sample); otherwise this edge falls back to global gathers;
//   C  every (pixel, channel group) lane walks its 8 planes: 4 ds_read_b128 taps, bilinear weights with
//      padding-zero semantics, sum / sum-of-squares in registers (edge order => deterministic).
// ---------------------------------------------------------------------------------------------------
constexpr int kWT = 8;            // pixel tile is kWT x kWT
constexpr int kWDB = 4;           // depth planes per workgroup
constexpr int kWinCells = 192;    // LDS window budget in feature cells (x 128 B)
constexpr int kWinFloats = 4 * 32 * 65;   // window buffer, also reused as [4 planes][32 ch][64+1 px] out tile
constexpr int kWinMaxE = 16;      // edges per pipelined chunk

struct PsvWinParams {
  PsvParams b;
  int ntx, nty;
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__global__ __launch_bounds__(512, 4) void psv_variance_win_kernel(PsvWinParams pp) {
  constexpr int C = 32;
  const PsvParams& p = pp.b;
  // double-buffered feature window (buffer 0 is reused as the [4 planes][32 ch][64+1 px] output tile),
  // triple-buffered sample positions / bounding boxes: edge e+1 is projected and its window prefetched
  // into registers while edge e is being sampled -> ONE barrier per edge, global latency hidden.
  __shared__ __attribute__((aligned(16))) float s_win[2][kWinFloats];
  __shared__ float2 s_ixy[3][kWDB][64];
  __shared__ int s_bbox[3][4];     // xmin, ymin, xmax, ymax of floor(ix), floor(iy) over valid samples
  __shared__ float s_ref[24];
  __shared__ float s_P[kWinMaxE][12];
  __shared__ int s_base[kWinMaxE];

  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % pp.ntx; b /= pp.ntx;
  const int ty = b % pp.nty; b /= pp.nty;
  const int n_dchunk = (p.D + kWDB - 1) / kWDB;
  const int dchunk = b % n_dchunk;
  const int r = b / n_dchunk;
  const int P = p.h * p.w;
  const int e_begin = p.edge_ofs[r], ne = p.edge_ofs[r + 1] - e_begin;
  const int ref = p.ref_img[r];
  const int d_first = dchunk * kWDB;
  const int nd = min(kWDB, p.D - d_first);

  if (tid == 0) {
    const float* Kp = p.K + ref * 9;
    double a = Kp[0], bb = Kp[1], c = Kp[2], d = Kp[3], e = Kp[4], f = Kp[5], g = Kp[6], hh = Kp[7], i = Kp[8];
    double det = a * (e * i - f * hh) - bb * (d * i - f * g) + c * (d * hh - e * g), id = 1.0 / det;
    s_ref[0] = (float)((e * i - f * hh) * id); s_ref[1] = (float)((c * hh - bb * i) * id); s_ref[2] = (float)((bb * f - c * e) * id);
    s_ref[3] = (float)((f * g - d * i) * id);  s_ref[4] = (float)((a * i - c * g) * id);   s_ref[5] = (float)((c * d - a * f) * id);
    s_ref[6] = (float)((d * hh - e * g) * id); s_ref[7] = (float)((bb * g - a * hh) * id); s_ref[8] = (float)((a * e - bb * d) * id);
  }
  if (tid >= 64 && tid < 73) s_ref[9 + tid - 64] = p.R[ref * 9 + tid - 64];
  if (tid >= 128 && tid < 131) s_ref[18 + tid - 128] = p.t[ref * 3 + tid - 128];
  __syncthreads();

  // ---- projection role: sample (plane sd, pixel sp) -------------------------------------------------
  const int sd = tid >> 6, sp = tid & 63;
  const int sgx = tx * kWT + (sp & 7), sgy = ty * kWT + (sp >> 3);
  const bool projector = sd < kWDB;            // waves beyond kWDB*64 samples only stage and sample
  const bool s_ok = projector && sd < nd && sgx < p.w && sgy < p.h;
  float X, Y, Z;
  {
    const float xf = (p.w > 1 && sgx == p.w - 1) ? (float)(p.W - 1) : (float)((double)sgx * p.x_step);
    const float yf = (p.h > 1 && sgy == p.h - 1) ? (float)(p.H - 1) : (float)((double)sgy * p.y_step);
    const int d = d_first + sd;
    const float z = (d == p.D - 1 && p.D > 1) ? (float)p.z_end : (float)(p.z_start + (double)d * p.z_step);
    const float p0 = xf * z, p1 = yf * z, p2 = z;
    const float c0 = s_ref[0] * p0 + s_ref[1] * p1 + s_ref[2] * p2 - s_ref[18];
    const float c1 = s_ref[3] * p0 + s_ref[4] * p1 + s_ref[5] * p2 - s_ref[19];
    const float c2 = s_ref[6] * p0 + s_ref[7] * p1 + s_ref[8] * p2 - s_ref[20];
    X = s_ref[9] * c0 + s_ref[12] * c1 + s_ref[15] * c2;
    Y = s_ref[10] * c0 + s_ref[13] * c1 + s_ref[16] * c2;
    Z = s_ref[11] * c0 + s_ref[14] * c1 + s_ref[17] * c2;
  }
  // ---- sampling role: pixel cp, channel group cg ------------------------------------------------------
  const int cp = tid >> 3, cg = tid & 7;
  const float Wm1 = (float)(p.W - 1), Hm1 = (float)(p.H - 1);
  const float Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);
  const float kNaN = __int_as_float(0x7fc00000);

  float acc_s[kWDB][4], acc_q[kWDB][4];
#pragma unroll
  for (int d = 0; d < kWDB; ++d)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc_s[d][k] = acc_q[d][k] = 0.f;

  struct Win { int x0, y0, w, h; bool nonempty, lds; };

  // A: project this thread's sample for edge e (chunk-local), reduce the cell bounding box
  auto project = [&](int e, int buf) {
    if (!projector) return;                      // wave-uniform
    const float* Pm = s_P[e];
    const float qx = Pm[0] * X + Pm[1] * Y + Pm[2] * Z + Pm[3];
    const float qy = Pm[4] * X + Pm[5] * Y + Pm[6] * Z + Pm[7];
    const float qz = Pm[8] * X + Pm[9] * Y + Pm[10] * Z + Pm[11];
    const float zb = fabsf(qz) + 1e-8f;
    const float u = qx / zb, v = qy / zb;
    const float gx = (u / Wm1) * 2.f - 1.f, gy = (v / Hm1) * 2.f - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * Wfm1, iy = ((gy + 1.f) / 2.f) * Hfm1;
    const bool any = s_ok && (ix > -1.f) && (ix < Wfm1 + 1.f) && (iy > -1.f) && (iy < Hfm1 + 1.f);
    s_ixy[buf][sd][sp] = any ? make_float2(ix, iy) : make_float2(kNaN, kNaN);
    const int fx = (int)floorf(any ? ix : 0.f), fy = (int)floorf(any ? iy : 0.f);
    const int x0 = wave_min_i(any ? fx : 0x7fffffff), y0 = wave_min_i(any ? fy : 0x7fffffff);
    const int x1 = wave_max_i(any ? fx : -0x7fffffff), y1 = wave_max_i(any ? fy : -0x7fffffff);
    if ((tid & 63) == 0 && x1 >= x0) {
      atomicMin(&s_bbox[buf][0], x0); atomicMin(&s_bbox[buf][1], y0);
      atomicMax(&s_bbox[buf][2], x1); atomicMax(&s_bbox[buf][3], y1);
    }
  };
  auto window = [&](int buf) {
    Win wn;
    wn.nonempty = s_bbox[buf][2] >= s_bbox[buf][0];
    wn.x0 = wn.nonempty ? max(s_bbox[buf][0], 0) : 0;
    wn.y0 = wn.nonempty ? max(s_bbox[buf][1], 0) : 0;
    const int x1 = wn.nonempty ? min(s_bbox[buf][2] + 1, p.Wf - 1) : 0, y1 = wn.nonempty ? min(s_bbox[buf][3] + 1, p.Hf - 1) : 0;
    wn.w = x1 - wn.x0 + 1; wn.h = y1 - wn.y0 + 1;
    wn.lds = wn.nonempty && wn.w * wn.h <= kWinCells;
    return wn;
  };
  constexpr int kStg = kWinCells * 8 / 512;      // float4 per thread for a full window
  static_assert(kWinCells * 8 % 512 == 0 && kWDB % 4 == 0 && kWDB * 64 <= 512, "window kernel geometry");
  float4 stg[kStg];
  auto prefetch = [&](int e, const Win& wn) {    // B, first half: window cells -> registers
    if (!wn.lds) return;
    const float* fimg = p.featT + (size_t)s_base[e] * C;
    const int nf4 = wn.w * wn.h * 8;
#pragma unroll
    for (int k = 0; k < kStg; ++k) {
      const int i = tid + k * 512;
      if (i < nf4) {
        const int cell = i >> 3, c4 = (i & 7) * 4;
        const int cy = cell / wn.w, cx = cell - cy * wn.w;
        stg[k] = *reinterpret_cast<const float4*>(fimg + ((wn.y0 + cy) * p.Wf + wn.x0 + cx) * C + c4);
      }
    }
  };
  auto commit = [&](int wbuf, const Win& wn) {   // B, second half: registers -> LDS window
    if (!wn.lds) return;
    const int nf4 = wn.w * wn.h * 8;
#pragma unroll
    for (int k = 0; k < kStg; ++k) {
      const int i = tid + k * 512;
      if (i < nf4) *reinterpret_cast<float4*>(&s_win[wbuf][(i >> 3) * C + (i & 7) * 4]) = stg[k];
    }
  };

  for (int ec = 0; ec < ne; ec += kWinMaxE) {
    const int nec = min(kWinMaxE, ne - ec);
    __syncthreads();
    if (tid < nec * 12) {
      int e = tid / 12, ij = tid % 12, i = ij / 4, j = ij % 4;
      int src = p.edge_src[e_begin + ec + e];
      const float* Kp = p.K + src * 9; const float* Rp = p.R + src * 9; const float* tp = p.t + src * 3;
      float v;
      if (j < 3) v = Kp[i * 3 + 0] * Rp[0 * 3 + j] + Kp[i * 3 + 1] * Rp[1 * 3 + j] + Kp[i * 3 + 2] * Rp[2 * 3 + j];
      else v = Kp[i * 3 + 0] * tp[0] + Kp[i * 3 + 1] * tp[1] + Kp[i * 3 + 2] * tp[2];
      s_P[e][ij] = v;
      if (ij == 0) s_base[e] = src * p.Hf * p.Wf;
    }
    if (tid >= 256 && tid < 256 + 12) s_bbox[(tid - 256) >> 2][(tid - 256) & 3] = ((tid & 3) < 2) ? 0x7fffffff : -0x7fffffff;
    __syncthreads();
    project(0, 0);
    __syncthreads();
    Win cur = window(0);
    prefetch(0, cur);
    for (int e = 0; e < nec; ++e) {
      const int ib = e % 3, wb = e & 1;
      commit(wb, cur);
      if (tid < 4) s_bbox[(e + 2) % 3][tid] = (tid < 2) ? 0x7fffffff : -0x7fffffff;   // free since edge e-1
      const bool more = e + 1 < nec;
      if (more) project(e + 1, (e + 1) % 3);
      __syncthreads();
      Win nxt = cur;
      if (more) { nxt = window((e + 1) % 3); prefetch(e + 1, nxt); }
      // ---- C: sample edge e ---------------------------------------------------------------------------
      if (cur.nonempty) {
        const float* fimg = p.featT + (size_t)s_base[e] * C + cg * 4;
        const float* wbase = &s_win[wb][cg * 4];
#pragma unroll
        for (int dd = 0; dd < kWDB; ++dd) {
          const float2 q = s_ixy[ib][dd][cp];
          const float ix = q.x, iy = q.y;
          if (ix == ix) {
            const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.f, y1 = y0 + 1.f;
            const bool vx0 = x0 >= 0.f, vx1 = x1 <= Wfm1, vy0 = y0 >= 0.f, vy1 = y1 <= Hfm1;
            const float w00 = (vx0 && vy0) ? (x1 - ix) * (y1 - iy) : 0.f, w01 = (vx1 && vy0) ? (ix - x0) * (y1 - iy) : 0.f;
            const float w10 = (vx0 && vy1) ? (x1 - ix) * (iy - y0) : 0.f, w11 = (vx1 && vy1) ? (ix - x0) * (iy - y0) : 0.f;
            const int xi0 = vx0 ? (int)x0 : 0, xi1 = vx1 ? (int)x1 : 0, yi0 = vy0 ? (int)y0 : 0, yi1 = vy1 ? (int)y1 : 0;
            float4 v00, v01, v10, v11;
            if (cur.lds) {
              // clamped coordinates lie inside the window by construction of the bounding box
              const int ax0 = max(xi0 - cur.x0, 0), ax1 = max(xi1 - cur.x0, 0);
              const int ay0 = max(yi0 - cur.y0, 0), ay1 = max(yi1 - cur.y0, 0);
              v00 = *reinterpret_cast<const float4*>(wbase + (ay0 * cur.w + ax0) * C);
              v01 = *reinterpret_cast<const float4*>(wbase + (ay0 * cur.w + ax1) * C);
              v10 = *reinterpret_cast<const float4*>(wbase + (ay1 * cur.w + ax0) * C);
              v11 = *reinterpret_cast<const float4*>(wbase + (ay1 * cur.w + ax1) * C);
            } else {
              v00 = *reinterpret_cast<const float4*>(fimg + (yi0 * p.Wf + xi0) * C);
              v01 = *reinterpret_cast<const float4*>(fimg + (yi0 * p.Wf + xi1) * C);
              v10 = *reinterpret_cast<const float4*>(fimg + (yi1 * p.Wf + xi0) * C);
              v11 = *reinterpret_cast<const float4*>(fimg + (yi1 * p.Wf + xi1) * C);
            }
            float4 s;
            s.x = v00.x * w00; s.y = v00.y * w00; s.z = v00.z * w00; s.w = v00.w * w00;
            s.x += v01.x * w01; s.y += v01.y * w01; s.z += v01.z * w01; s.w += v01.w * w01;
            s.x += v10.x * w10; s.y += v10.y * w10; s.z += v10.z * w10; s.w += v10.w * w10;
            s.x += v11.x * w11; s.y += v11.y * w11; s.z += v11.z * w11; s.w += v11.w * w11;
            acc_s[dd][0] += s.x; acc_s[dd][1] += s.y; acc_s[dd][2] += s.z; acc_s[dd][3] += s.w;
            acc_q[dd][0] += s.x * s.x; acc_q[dd][1] += s.y * s.y; acc_q[dd][2] += s.z * s.z; acc_q[dd][3] += s.w * s.w;
          }
        }
      }
      cur = nxt;
    }
  }
  // ---- variance -> LDS transpose -> store, 4 planes at a time ---------------------------------------------
  const float cnt = (float)max(ne, 1);
  float (*s_out)[C][65] = reinterpret_cast<float (*)[C][65]>(&s_win[0][0]);
#pragma unroll
  for (int half = 0; half < kWDB / 4; ++half) {
    __syncthreads();
#pragma unroll
    for (int d4 = 0; d4 < 4; ++d4) {
      const int dd = half * 4 + d4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float avg = acc_s[dd][k] / cnt, avg_sq = acc_q[dd][k] / cnt;
        s_out[d4][cg * 4 + k][cp] = __fsub_rn(avg_sq, __fmul_rn(avg, avg));
      }
    }
    __syncthreads();
    const int px = tid & 63;
    const int gx = tx * kWT + (px & 7), gy = ty * kWT + (px >> 3);
    if (gx < p.w && gy < p.h) {
      for (int dc = tid >> 6; dc < 4 * C; dc += 8) {
        const int d4 = dc / C, c = dc % C, d = d_first + half * 4 + d4;
        if (d < p.D) p.var[(((size_t)r * C + c) * p.D + d) * P + gy * p.w + gx] = s_out[d4][c][px];
      }
    }
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

extern "C" size_t v3d_psv_workspace_bytes(int n_img, int C, int Hf, int Wf) {
  return v3d::align_up((size_t)n_img * C * Hf * Wf * sizeof(float), 256);
}

extern "C" int v3d_psv_variance_f32(const float* feat, const float* K, const float* R,
                                    const float* t, const int32_t* ref_img,
                                    const int32_t* edge_ofs, const int32_t* edge_src, int n_img,
                                    int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                                    double depth_start, double depth_interval, int D, int h, int w,
                                    float* var, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  V3D_REQUIRE(feat && K && R && t && ref_img && edge_ofs && edge_src && var && workspace,
              V3D_ERR_BAD_ARG, "v3d_psv_variance_f32: null pointer argument");
  V3D_REQUIRE(C == 32 || C == 16, V3D_ERR_UNSUPPORTED,
              "v3d_psv_variance_f32: C=%d unsupported (16 or 32)", C);
  V3D_REQUIRE(n_img > 0 && n_ref > 0 && n_edges >= 0 && Hf > 0 && Wf > 0 && H > 1 && W > 1 &&
                  D > 0 && h > 0 && w > 0,
              V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: bad shape");
  V3D_REQUIRE((size_t)n_img * Hf * Wf * C < (size_t)1 << 31, V3D_ERR_BAD_SHAPE,
              "v3d_psv_variance_f32: feature tensor exceeds 2^31 elements");
  V3D_REQUIRE(workspace_bytes >= v3d_psv_workspace_bytes(n_img, C, Hf, Wf),
              V3D_ERR_WORKSPACE_TOO_SMALL, "v3d_psv_variance_f32: workspace %zu < %zu",
              workspace_bytes, v3d_psv_workspace_bytes(n_img, C, Hf, Wf));
  hipStream_t s = (hipStream_t)stream;
  float* featT = (float*)workspace;
  v3d::transpose_channel_last(feat, featT, n_img, C, Hf * Wf, s);
  V3D_CHECK_LAUNCH("transpose_channel_last_kernel");

  PsvParams p;
  p.featT = featT; p.K = K; p.R = R; p.t = t;
  p.ref_img = ref_img; p.edge_ofs = edge_ofs; p.edge_src = edge_src; p.var = var;
  p.n_img = n_img; p.n_ref = n_ref; p.Hf = Hf; p.Wf = Wf; p.H = H; p.W = W; p.D = D;
  p.h = h; p.w = w;
  p.n_ptile = (h * w + kPix - 1) / kPix;
  p.x_step = w > 1 ? (double)(W - 1) / (double)(w - 1) : 0.0;
  p.y_step = h > 1 ? (double)(H - 1) / (double)(h - 1) : 0.0;
  const double depth_end = depth_start + (double)(D - 1) * depth_interval;
  p.z_start = depth_start;
  p.z_step = D > 1 ? (depth_end - depth_start) / (double)(D - 1) : 0.0;
  p.z_end = depth_end;
  const int n_dchunk = (D + kDB - 1) / kDB;
  const long long blocks = (long long)n_ref * n_dchunk * p.n_ptile;
  V3D_REQUIRE(blocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: grid too large");
  // The LDS-window kernel is correct (same parity tests) but currently slower than the gather kernel
  // (2.1-2.9 ms vs 1.3 ms per 32-view launch); it stays opt-in for further work.
  static const bool use_win = getenv("V3D_PSV_WINDOW") != nullptr;
  if (C == 32 && use_win) {
    PsvWinParams pw;
    pw.b = p;
    pw.ntx = (w + kWT - 1) / kWT; pw.nty = (h + kWT - 1) / kWT;
    const long long wblocks = (long long)n_ref * ((D + kWDB - 1) / kWDB) * pw.ntx * pw.nty;
    V3D_REQUIRE(wblocks < (1ll << 31), V3D_ERR_BAD_SHAPE, "v3d_psv_variance_f32: grid too large");
    v3d::TimedScope ts("psv_variance", s);
    psv_variance_win_kernel<<<(unsigned)wblocks, 512, 0, s>>>(pw);
  } else {
    v3d::TimedScope ts("psv_variance", s);
    if (C == 32) psv_variance_kernel<32><<<(unsigned)blocks, kThreads, 0, s>>>(p);
    else psv_variance_kernel<16><<<(unsigned)blocks, kThreads, 0, s>>>(p);
  }
  V3D_CHECK_LAUNCH("psv_variance_kernel");
  return V3D_OK;
}
