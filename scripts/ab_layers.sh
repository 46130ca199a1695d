# usage: bash scripts/ab_layers.sh  -- sweeps tile configs of the memory-bound conv layers
run() { # tag, flags
  touch 3dvnet_amd/csrc/costreg.hip
  V3D_EXTRA_FLAGS="$2" python 3dvnet_amd/build.py > /dev/null 2>&1 && python scripts/bench_layers.py --tag "$1" 2>&1 | tail -1 | sed 's/transpose[^|]*psv_variance=[0-9.]* //' | cut -c1-200 || echo "$1 FAILED"
}
run base ""
run L9_4x8x56_o2 "-DV3D_L9_CFG=kDeconvS2,16,8,4,8,56,8,2"
run L9_2x8x28_o4 "-DV3D_L9_CFG=kDeconvS2,16,8,2,8,28,8,4"
run L9_4x4x28_o4 "-DV3D_L9_CFG=kDeconvS2,16,8,4,4,28,8,4"
run L9_8x8x28_o2 "-DV3D_L9_CFG=kDeconvS2,16,8,8,8,28,8,2"
run L9_4x8x28_ck16 "-DV3D_L9_CFG=kDeconvS2,16,8,4,8,28,16,3"
run L1_2x8x28 "-DV3D_L1_CFG=kConvS2,8,16,2,8,28,4"
run L1_4x4x28 "-DV3D_L1_CFG=kConvS2,8,16,4,4,28,4"
run L1_2x4x28_ck8 "-DV3D_L1_CFG=kConvS2,8,16,2,4,28,8"
run L2_4x8x28 "-DV3D_L2_CFG=kConvS1,16,16,4,8,28,8"
run L2_4x4x28_ck16 "-DV3D_L2_CFG=kConvS1,16,16,4,4,28,16"
run L8_4x8x28 "-DV3D_L8_CFG=kDeconvS2,32,16,4,8,28,8"
run L8_4x14x28_o3 "-DV3D_L8_CFG=kDeconvS2,32,16,4,14,28,8,3"
touch 3dvnet_amd/csrc/costreg.hip; python 3dvnet_amd/build.py > /dev/null
