# Developer: the backbone leg of bench.py alone (V3D_LIB_OVERRIDE=<lib> for an A/B against another build)
python - <<'PY'
import importlib, os, sys, torch, json
sys.path.insert(0, '.')
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'): libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
import bench
print(json.dumps({k: v for k, v in bench.bench_backbone(torch.device('cuda:0')).items() if k in ('ms_per_batch', 'kernel_ms_per_batch', 'kernels', 'max_diff_vs_stock_modules_of_range')}))
PY
