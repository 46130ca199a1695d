# Developer: the sparse-convolution kernels side by side on the three level shapes of the cfg3 scene, then the whole U-Net forward
timeout 600 python -m pytest tests/test_parity_net_gpu.py -x -q -k "gather_gemm_rounds or sparse" 2>&1 | tail -3
for a in "13434 128" "2719 128" "59975 64"; do set -- $a; for o in gemm_pipe=0 gemm_pipe=1 gemm_pipe=2 gemm_pipe=1,gemm_round_rows=32 gemm_pipe=1,gemm_round_rows=64 gemm_pipe=1,gemm_round_rows=128; do echo -n "== $1 x $2 $o: "; V3D_OPTIONS=$o timeout 120 python scripts/phase_sparse_gemm.py --rows $1 --c $2 --absent 0.6 2>&1 | tail -1; done; done
timeout 300 python scripts/micro/unet_levels.py 2>&1 | tail -28
