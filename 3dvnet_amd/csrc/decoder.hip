// Row C2b tail + C3 of SURVEY.md §8a: the hypothesis decoder's last Conv1d(h_dim -> 1, k3, pad 1, bias)
// along the hypothesis axis, softmax over the hypotheses (refinement.py:24,43) and, optionally, the
// expected depth offset sum_i p_i * vals_i (lightningmodel.py:238-241).  One wave per point.
#include "v3d_common.h"

namespace {

constexpr int kMaxHyp = 16;

__global__ __launch_bounds__(256) void decoder_head_kernel(const float* __restrict__ act, int n_pts,
                                                           int n_hyp, int C, const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ vals,
                                                           float* __restrict__ preds,
                                                           float* __restrict__ expect) {
  const int lane = threadIdx.x & 63;
  const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= n_pts) return;
  float score[kMaxHyp];
#pragma unroll
  for (int h = 0; h < kMaxHyp; ++h) score[h] = 0.f;
  const float* a = act + (size_t)pt * n_hyp * C;
  for (int c = lane; c < C; c += 64) {
    const float w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];   // weight [1, C, 3]
#pragma unroll
    for (int h = 0; h < kMaxHyp; ++h) {
      if (h < n_hyp) {
        const float x = a[(size_t)h * C + c];
        // out[h'] = sum_t in[h' + t - 1] w[t]  =>  in[h] feeds out[h+1] (t=0), out[h] (t=1), out[h-1] (t=2)
        if (h + 1 < n_hyp) score[h + 1] += x * w0;
        score[h] += x * w1;
        if (h > 0) score[h - 1] += x * w2;
      }
    }
  }
#pragma unroll
  for (int h = 0; h < kMaxHyp; ++h) {
    float v = score[h];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    score[h] = v + bias[0];
  }
  if (lane == 0) {
    float m = -INFINITY;
    for (int h = 0; h < n_hyp; ++h) m = fmaxf(m, score[h]);
    float s = 0.f;
    for (int h = 0; h < n_hyp; ++h) s += expf(score[h] - m);
    float e = 0.f;
    for (int h = 0; h < n_hyp; ++h) {
      const float pr = expf(score[h] - m) / s;
      preds[(size_t)pt * n_hyp + h] = pr;
      if (vals) e += vals[h] * pr;
    }
    if (expect) expect[pt] = e;
  }
}

}  // namespace

extern "C" int v3d_decoder_head_f32(const float* act, int n_pts, int n_hyp, int C, const float* weight,
                                    const float* bias, const float* offset_vals, float* preds,
                                    float* expect, void* stream) {
  V3D_REQUIRE(act && weight && bias && preds, V3D_ERR_BAD_ARG, "v3d_decoder_head_f32: null argument");
  V3D_REQUIRE(n_hyp >= 1 && n_hyp <= kMaxHyp && C >= 1 && n_pts >= 0, V3D_ERR_BAD_SHAPE,
              "v3d_decoder_head_f32: bad shape (n_hyp=%d)", n_hyp);
  V3D_REQUIRE(!expect || offset_vals, V3D_ERR_BAD_ARG, "v3d_decoder_head_f32: expect without offset_vals");
  if (n_pts == 0) return V3D_OK;
  hipStream_t s = (hipStream_t)stream;
  v3d::TimedScope ts("decoder_head", s);
  decoder_head_kernel<<<(n_pts + 3) / 4, 256, 0, s>>>(act, n_pts, n_hyp, C, weight, bias, offset_vals, preds, expect);
  V3D_CHECK_LAUNCH("decoder_head_kernel");
  return V3D_OK;
}
