mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_gp.py tests/test_gpu_gp_chunked.py tests/test_gpu_gp_lane.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_timed_config.py -x -q -m gpu 2>&1 | tail -5
python - <<'PY' > gpurun_out/r3m/gp_cond_leg.json 2>gpurun_out/r3m/gp_cond_leg.err
import json, torch, sys
sys.path.insert(0, '.')
import bench, exoplanet_amd as xo
from exoplanet_amd import ops
print(json.dumps(bench.extra_gp_conditioning(xo, ops, torch.device('cuda:0'), 1024)))
PY
cat gpurun_out/r3m/gp_cond_leg.json; tail -3 gpurun_out/r3m/gp_cond_leg.err
