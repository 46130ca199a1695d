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
  // Conv1d(k3) layers with 128 outputs (the hypothesis decoder, n_seg = 3, N = 128, K % 16 == 0) carry a third image for the
  // fused decoder's 32x32x16 matrix instructions at dec_ofs (0 = absent): per 16-wide K step one 24 KB slab
  // [3 taps][hi, lo][4 row blocks of 32 outputs][64 lanes][4 words]; lane l = (g = l >> 5, i = l & 31) holds output 32 mb + i and
  // the inputs c(e) = 16 step + 8 (e >> 2) + 4 g + (e & 3), e = 0..7 -- the order in which a lane of the PREVIOUS layer's
  // accumulator tile holds its column's channels, so a layer's output registers are the next layer's B fragments.
  size_t dec_ofs;
  int has_bias, has_gn;
};
