# Developer: PointNet's dense 200 k-row layers on the one-step kernel (default) and the rounds kernel (gemm_rounds=2): cfg3 scene time
# and the per-name kernel totals
for o in gemm_rounds=1 gemm_rounds=2; do echo "== $o"; V3D_OPTIONS=$o python bench.py --config cfg3 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:(v['total_ms'],v['launches']) for k,v in d['kernels'].items() if 'gemm' in k or 'linear' in k})"; done
