# Developer: ablations of the sparse-convolution pipeline kernel (0 full, 1 no MFMAs, 2 no fragment loads, 3 no gathers, 4 no split /
# LDS writes) with the phase counters of both roles (0 prologue, 1 fragment issue, 2 matrix phase, 3 barrier | loader: 4 commit incl. the
# wait for the gathers, 5 gather issue, 6 barrier, 7 prologue)
for a in 0 1 2 3 4; do for cfgs in "13434 128 gemm_pipe=1,gemm_round_rows=64" "13434 128 gemm_pipe=3,gemm_round_rows=32" "59975 64 gemm_pipe=1,gemm_round_rows=64"; do set -- $cfgs
echo -n "ablate $a $3: "; V3D_OPTIONS=$3 V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_pipe_a$a.so python scripts/phase_sparse_gemm.py --rows $1 --c $2 --absent 0.6 2>&1 | tail -1; done; done
