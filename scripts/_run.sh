mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r6_pytest_gpu.txt
cat gpurun_out/r6_pytest_gpu.txt
timeout 900 python bench.py --config cfg3 --steps 6 --warmup 2 > gpurun_out/r6_bench_cfg3.json 2> gpurun_out/r6_bench_cfg3.err
tail -3 gpurun_out/r6_bench_cfg3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_cfg3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_fp32_exact','ms_per_step_fp32_exact')})
print(json.dumps(d['parity'],indent=1)[:3500])
PY
