# Developer: phase counters of the sparse-convolution pipeline kernel (0 prologue, 1 fragment issue, 2 matrix instructions incl.
# the wait for their fragments, 3 barrier | loader wave: 4 commit incl. the wait for the gathers, 5 gather issue, 6 barrier, 7 prologue)
for a in "13434 128 64" "13434 128 32" "2719 128 32" "59975 64 64"; do set -- $a
V3D_OPTIONS=gemm_round_rows=$3 V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_ggphase.so python scripts/phase_sparse_gemm.py --rows $1 --c $2 --absent 0.6 2>&1 | tail -1; done
