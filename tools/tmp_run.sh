cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/transit_stress.py 3 150 2>&1 | tail -2
timeout 300 python tools/ttv_stress.py 6 100 2>&1 | tail -2
for rep in 1 2 3; do
for v in base nofold; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$GRAFT_REPO_ROOT/tests/_build/variants/$v.so; fi
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras --no-stats 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))"
done
done
