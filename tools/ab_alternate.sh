# alternating A/B of library variants (clocks and boxes drift: only interleaved runs compare): tools/ab_alternate.sh <reps> <variant> ...
R=$GRAFT_REPO_ROOT; reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
    python $R/tools/ab_transit.py --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v', ' '.join('%s=%.1f'%(k[8:],v) for k,v in d.items() if k.startswith('c2_1024') and not k.endswith('check')))"
  done
done
