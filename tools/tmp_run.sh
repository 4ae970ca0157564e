cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in base oldwin; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$GRAFT_REPO_ROOT/tests/_build/variants/$v.so; fi
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras --no-stats 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))"
done
done
