mkdir -p gpurun_out/r3r
python -m pytest tests/test_gpu_gp.py tests/test_gpu_gp_chunked.py tests/test_gpu_gp_lane.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "^E  |^>|passed|failed" | head
python - <<'PY' > gpurun_out/r3r/gp_cond_leg.json 2>gpurun_out/r3r/gp_cond_leg.err
import json, torch, sys
sys.path.insert(0, '.')
import bench, exoplanet_amd as xo
from exoplanet_amd import ops
print(json.dumps(bench.extra_gp_conditioning(xo, ops, torch.device('cuda:0'), 1024)))
PY
python -c "
import json
d=json.load(open('gpurun_out/r3r/gp_cond_leg.json'))
print({k:(round(v['median_ms'],3), round(v.get('over_clean',0),2)) for k,v in d.items() if isinstance(v,dict) and 'median_ms' in v})"
