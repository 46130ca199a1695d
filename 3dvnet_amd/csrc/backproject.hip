// Rows B1-B2 and C1 of SURVEY.md §8a: back-project depth pixels (or 2n+1 depth hypotheses per pixel) to
// world points, re-project them into every source view of the reference, bilinearly sample the
// quarter-resolution features and reduce to the cross-view variance.  Reference semantics:
// mv3d/lightningmodel.py:132-174 (construct_feature_rich_pointcloud) and :187-235 (run_pointflow),
// mv3d/utils.py:67-83 (build_img_pts).
//
// One kernel, n_hyp = 1 (scene point cloud) or 2n+1 (point-flow hypotheses).  8 (C=32) lanes per
// sample, each owning 4 channels of the channel-last feature tensor; sums over edges in edge order.
// The problem is small (3 136 x n_hyp samples per view): latency-bound, so views are batched per launch.
#include "v3d_common.h"

namespace {

constexpr int kMaxE = 8;

struct BpParams {
  const float* depth;    // [n_ref, h*w]
  const float* featT;    // [n_img, Hf, Wf, C]
  const float* K; const float* R; const float* t;
  const int* ref_img; const int* edge_ofs; const int* edge_src;
  float* pts;            // [n_ref*P, n_hyp, 3]
  float* var;            // [n_ref*P, n_hyp, C]
  int n_ref, Hf, Wf, H, W, h, w, n_hyp, n_half;
  double x_step, y_step, offset;
  // launch constants from the host (round 4: per lane they were two IEEE f64 divisions and two run-time integer divisions
  // through the float reciprocal, all quarter- / half-rate instructions)
  float rWm1, rHm1;                 // (float)(1.0 / (double)(W - 1)), (float)(1.0 / (double)(H - 1))
  unsigned m_hyp, m_w;              // v3d::magic_u32() of n_hyp and w (0: divide)
};

template <int C>
__global__ __launch_bounds__(256) void backproject_variance_kernel(BpParams p) {
  constexpr int LP = C / 4;
  __shared__ float s_ref[24];
  __shared__ float s_P[kMaxE][12];
  __shared__ int s_base[kMaxE];
  const int tid = threadIdx.x;
  const int r = blockIdx.y;
  const int P = p.h * p.w;
  const int ref = p.ref_img[r];
  const int e_begin = p.edge_ofs[r], ne = p.edge_ofs[r + 1] - e_begin;

  if (tid == 0) {   // K^-1 (fp64 adjugate, rounded to f32; lightningmodel.py:138,191)
    const float* Kp = p.K + ref * 9;
    double a = Kp[0], bb = Kp[1], c = Kp[2], d = Kp[3], e = Kp[4], f = Kp[5], g = Kp[6], hh = Kp[7], i = Kp[8];
    double det = a * (e * i - f * hh) - bb * (d * i - f * g) + c * (d * hh - e * g), id = 1.0 / det;
    s_ref[0] = (float)((e * i - f * hh) * id); s_ref[1] = (float)((c * hh - bb * i) * id); s_ref[2] = (float)((bb * f - c * e) * id);
    s_ref[3] = (float)((f * g - d * i) * id);  s_ref[4] = (float)((a * i - c * g) * id);   s_ref[5] = (float)((c * d - a * f) * id);
    s_ref[6] = (float)((d * hh - e * g) * id); s_ref[7] = (float)((bb * g - a * hh) * id); s_ref[8] = (float)((a * e - bb * d) * id);
  }
  if (tid >= 64 && tid < 73) s_ref[9 + tid - 64] = p.R[ref * 9 + tid - 64];
  if (tid >= 128 && tid < 131) s_ref[18 + tid - 128] = p.t[ref * 3 + tid - 128];

  const int sample = (blockIdx.x * 256 + tid) / LP;       // (pixel, hypothesis) of this lane group
  const int cg = tid % LP;
  const bool active = sample < P * p.n_hyp;
  const unsigned spix = active ? v3d::udiv_magic((unsigned)sample, (unsigned)p.n_hyp, p.m_hyp) : 0u;
  const int pix = (int)spix, hyp = active ? sample - (int)spix * p.n_hyp : 0;
  const int gy = (int)v3d::udiv_magic(spix, (unsigned)p.w, p.m_w), gx = pix - gy * p.w;
  const float xf = (p.w > 1 && gx == p.w - 1) ? (float)(p.W - 1) : (float)((double)gx * p.x_step);
  const float yf = (p.h > 1 && gy == p.h - 1) ? (float)(p.H - 1) : (float)((double)gy * p.y_step);
  // hypothesis depth: depth + i * offset with i * offset evaluated in double then rounded (python float)
  const float dep = v3d::add_rn(p.depth[(size_t)r * P + pix], (float)((double)(hyp - p.n_half) * p.offset));
  const float Wm1 = (float)(p.W - 1), Hm1 = (float)(p.H - 1), Wfm1 = (float)(p.Wf - 1), Hfm1 = (float)(p.Hf - 1);
  const float rWm1 = p.rWm1, rHm1 = p.rHm1;
  float X = 0.f, Y = 0.f, Z = 0.f;
  float4 acc_s = make_float4(0.f, 0.f, 0.f, 0.f), acc_q = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int ec = 0; ec < ne; ec += kMaxE) {
    const int nec = min(kMaxE, ne - ec);
    __syncthreads();
    if (tid < nec * 12) {      // P = K [R|t] of the chunk's source views (lightningmodel.py:152-154)
      int e = tid / 12, ij = tid % 12, i = ij / 4, j = ij % 4;
      int src = p.edge_src[e_begin + ec + e];
      const float* Kp = p.K + src * 9; const float* Rp = p.R + src * 9; const float* tp = p.t + src * 3;
      float v;
      // small batched product of torch.bmm: rounded products, sequential additions, no FMA (see psv_variance.hip)
      const float b0 = j < 3 ? Rp[0 * 3 + j] : tp[0], b1 = j < 3 ? Rp[1 * 3 + j] : tp[1], b2 = j < 3 ? Rp[2 * 3 + j] : tp[2];
      v = v3d::add_rn(v3d::add_rn(v3d::mul_rn(Kp[i * 3 + 0], b0), v3d::mul_rn(Kp[i * 3 + 1], b1)), v3d::mul_rn(Kp[i * 3 + 2], b2));
      s_P[e][ij] = v;
      if (ij == 0) s_base[e] = src * p.Hf * p.Wf;
    }
    __syncthreads();
    if (ec == 0) v3d::world_point(s_ref, xf, yf, dep, X, Y, Z);   // (lightningmodel.py:142-144,202-204)
    // The LP lanes of a sample need the same projection for every edge.  Lane cg projects edges cg, cg + LP, ... of the chunk
    // and the group reads the tap weights / offsets of edge e from lane e % LP (ds_bpermute): the pinned projection chain
    // (~80 VALU instructions per edge) runs once per sample instead of LP times.  (Lanes of an inactive sample run along with
    // pixel 0: the shuffles are wave-wide.)
    constexpr int kEPL = (kMaxE + LP - 1) / LP;                   // edges a lane projects per chunk
    float tw[kEPL][4];
    int to[kEPL][4];
#pragma unroll
    for (int q = 0; q < kEPL; ++q) {
      const int e = cg + q * LP;
#pragma unroll
      for (int k = 0; k < 4; ++k) { tw[q][k] = 0.f; to[q][k] = k == 0 ? -1 : 0; }      // offset 0 of tap 0 < 0: sample outside the image
      if (e < nec) {
        float ix, iy;
        v3d::sample_position(s_P[e], X, Y, Z, Wm1, rWm1, Hm1, rHm1, Wfm1, Hfm1, ix, iy);
        if (ix > -1.f && ix < Wfm1 + 1.f && iy > -1.f && iy < Hfm1 + 1.f) {
          const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.f, y1 = y0 + 1.f;
          const bool vx0 = x0 >= 0.f, vx1 = x1 <= Wfm1, vy0 = y0 >= 0.f, vy1 = y1 <= Hfm1;
          tw[q][0] = (vx0 && vy0) ? (x1 - ix) * (y1 - iy) : 0.f; tw[q][1] = (vx1 && vy0) ? (ix - x0) * (y1 - iy) : 0.f;
          tw[q][2] = (vx0 && vy1) ? (x1 - ix) * (iy - y0) : 0.f; tw[q][3] = (vx1 && vy1) ? (ix - x0) * (iy - y0) : 0.f;
          const int xi0 = vx0 ? (int)x0 : 0, xi1 = vx1 ? (int)x1 : 0, yi0 = vy0 ? (int)y0 : 0, yi1 = vy1 ? (int)y1 : 0;
          const int base = s_base[e];
          to[q][0] = (base + yi0 * p.Wf + xi0) * C; to[q][1] = (base + yi0 * p.Wf + xi1) * C;
          to[q][2] = (base + yi1 * p.Wf + xi0) * C; to[q][3] = (base + yi1 * p.Wf + xi1) * C;
        }
      }
    }
    const int lane0 = (tid & 63) & ~(LP - 1);
    const float* const fb = p.featT + cg * 4;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) {
      if (e < nec) {                                              // wave-uniform
        float w[4];
        int o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          w[k] = __shfl(tw[e / LP][k], lane0 + e % LP, 64);
          o[k] = __shfl(to[e / LP][k], lane0 + e % LP, 64);
        }
        const bool inside = o[0] >= 0;                            // outside: nothing is sampled (grid_sample's zero padding)
        const float4 v00 = *reinterpret_cast<const float4*>(fb + max(o[0], 0));
        const float4 v01 = *reinterpret_cast<const float4*>(fb + o[1]);
        const float4 v10 = *reinterpret_cast<const float4*>(fb + o[2]);
        const float4 v11 = *reinterpret_cast<const float4*>(fb + o[3]);
        // F.grid_sample's tap order as an FMA chain: ((nw + ne) + sw) + se (same sequence as the warp kernels)
        float4 s;
        s.x = v3d::mul_rn(v00.x, w[0]); s.y = v3d::mul_rn(v00.y, w[0]); s.z = v3d::mul_rn(v00.z, w[0]); s.w = v3d::mul_rn(v00.w, w[0]);
        s.x = __builtin_fmaf(v01.x, w[1], s.x); s.y = __builtin_fmaf(v01.y, w[1], s.y);
        s.z = __builtin_fmaf(v01.z, w[1], s.z); s.w = __builtin_fmaf(v01.w, w[1], s.w);
        s.x = __builtin_fmaf(v10.x, w[2], s.x); s.y = __builtin_fmaf(v10.y, w[2], s.y);
        s.z = __builtin_fmaf(v10.z, w[2], s.z); s.w = __builtin_fmaf(v10.w, w[2], s.w);
        s.x = __builtin_fmaf(v11.x, w[3], s.x); s.y = __builtin_fmaf(v11.y, w[3], s.y);
        s.z = __builtin_fmaf(v11.z, w[3], s.z); s.w = __builtin_fmaf(v11.w, w[3], s.w);
        if (!inside) s = make_float4(0.f, 0.f, 0.f, 0.f);
        acc_s.x += s.x; acc_s.y += s.y; acc_s.z += s.z; acc_s.w += s.w;
        acc_q.x = __builtin_fmaf(s.x, s.x, acc_q.x); acc_q.y = __builtin_fmaf(s.y, s.y, acc_q.y);
        acc_q.z = __builtin_fmaf(s.z, s.z, acc_q.z); acc_q.w = __builtin_fmaf(s.w, s.w, acc_q.w);
      }
    }
  }
  if (!active) return;
  const size_t row = ((size_t)r * P + pix) * p.n_hyp + hyp;
  if (cg == 0) { p.pts[row * 3] = X; p.pts[row * 3 + 1] = Y; p.pts[row * 3 + 2] = Z; }
  const float cnt = (float)max(ne, 1);
  // x / cnt is an IEEE division (~10 instructions, eight of them per lane); for a power-of-two count x * (1 / cnt) is the same
  // number exactly (block-uniform choice, as in the warp kernels)
  const bool cnt_pow2 = (max(ne, 1) & (max(ne, 1) - 1)) == 0;
  const float cnt_inv = 1.f / cnt;
  auto mean = [&](float x) __attribute__((always_inline)) { return cnt_pow2 ? x * cnt_inv : x / cnt; };
  float4 o;
  { float a = mean(acc_s.x), q = mean(acc_q.x); o.x = v3d::sub_rn(q, v3d::mul_rn(a, a)); }
  { float a = mean(acc_s.y), q = mean(acc_q.y); o.y = v3d::sub_rn(q, v3d::mul_rn(a, a)); }
  { float a = mean(acc_s.z), q = mean(acc_q.z); o.z = v3d::sub_rn(q, v3d::mul_rn(a, a)); }
  { float a = mean(acc_s.w), q = mean(acc_q.w); o.w = v3d::sub_rn(q, v3d::mul_rn(a, a)); }
  *reinterpret_cast<float4*>(p.var + row * C + cg * 4) = o;
}

}  // namespace

extern "C" size_t v3d_backproject_workspace_bytes(int n_img, int C, int Hf, int Wf) {
  return v3d::align_up((size_t)n_img * C * Hf * Wf * sizeof(float), 256);
}

extern "C" int v3d_backproject_variance_f32(const float* depth, const float* feat, const float* K,
                                            const float* R, const float* t, const int32_t* ref_img,
                                            const int32_t* edge_ofs, const int32_t* edge_src, int n_img,
                                            int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                                            int h, int w, double offset, int n_half, float* pts,
                                            float* var, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  // feat == NULL: `workspace` still holds the channel-last copy a previous call made of the same feature tensor (the scene
  // driver back-projects against the same 46 MB of features eight times per scene: 23 us of transposition per call)
  V3D_REQUIRE(depth && K && R && t && ref_img && edge_ofs && edge_src && pts && var && workspace,
              V3D_ERR_BAD_ARG, "v3d_backproject_variance_f32: null pointer argument");
  V3D_REQUIRE(C == 32 || C == 16, V3D_ERR_UNSUPPORTED, "v3d_backproject_variance_f32: C=%d unsupported", C);
  V3D_REQUIRE((long long)n_img * Hf * Wf * C < (1ll << 31), V3D_ERR_BAD_SHAPE,
              "v3d_backproject_variance_f32: feature tensor too large for 32-bit tap offsets");
  V3D_REQUIRE(n_img > 0 && n_ref > 0 && n_edges >= 0 && H > 1 && W > 1 && h > 0 && w > 0 && n_half >= 0,
              V3D_ERR_BAD_SHAPE, "v3d_backproject_variance_f32: bad shape");
  V3D_REQUIRE(workspace_bytes >= v3d_backproject_workspace_bytes(n_img, C, Hf, Wf),
              V3D_ERR_WORKSPACE_TOO_SMALL, "v3d_backproject_variance_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* featT = (float*)workspace;
  if (feat) {
    v3d::transpose_channel_last(feat, featT, n_img, C, Hf * Wf, s);
    V3D_CHECK_LAUNCH("transpose_channel_last_kernel");
  }
  BpParams p;
  p.depth = depth; p.featT = featT; p.K = K; p.R = R; p.t = t;
  p.ref_img = ref_img; p.edge_ofs = edge_ofs; p.edge_src = edge_src; p.pts = pts; p.var = var;
  p.n_ref = n_ref; p.Hf = Hf; p.Wf = Wf; p.H = H; p.W = W; p.h = h; p.w = w;
  p.n_hyp = 2 * n_half + 1; p.n_half = n_half; p.offset = offset;
  p.x_step = w > 1 ? (double)(W - 1) / (double)(w - 1) : 0.0;
  p.y_step = h > 1 ? (double)(H - 1) / (double)(h - 1) : 0.0;
  p.rWm1 = (float)(1.0 / (double)(W - 1));
  p.rHm1 = (float)(1.0 / (double)(H - 1));
  p.m_hyp = v3d::magic_u32((unsigned long long)h * w * p.n_hyp + 256, (unsigned)p.n_hyp);
  p.m_w = v3d::magic_u32((unsigned long long)h * w + 256, (unsigned)w);
  const long long lanes = (long long)h * w * p.n_hyp * (C / 4);
  dim3 grid((unsigned)((lanes + 255) / 256), n_ref);
  {
    v3d::TimedScope ts("backproject_variance", s);
    if (C == 32) backproject_variance_kernel<32><<<grid, 256, 0, s>>>(p);
    else backproject_variance_kernel<16><<<grid, 256, 0, s>>>(p);
  }
  V3D_CHECK_LAUNCH("backproject_variance_kernel");
  return V3D_OK;
}
