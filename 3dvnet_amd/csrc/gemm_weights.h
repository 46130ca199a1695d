// Packed weight image of one gather-GEMM layer (v3d_gemm_pack in gemm_gather.hip); also read by the fused hypothesis decoder
// (decoder.hip).
#pragma once
#include <cstddef>

struct v3d_gemm_weights {
  int N, K, KP, n_seg, MBW;
  float* dev;       // exact-fp32 fragments, then bias[N] (0 if none), gn_w[N], gn_b[N], then the split-bf16 image at bf_ofs:
                    // per (segment, 32-wide K chunk): [hi, lo][MB = 4 MBW][64 lanes][4 words]; lane l holds output channel
                    // mb*16 + (l & 15), k = chunk*32 + 8*(l >> 4) + e (e = 0..7, two bf16 per word)
  size_t bias_ofs, gnw_ofs, gnb_ofs, bf_ofs;
  int has_bias, has_gn;
};
