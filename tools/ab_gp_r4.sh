#!/bin/bash
# A/B of the GP configs (C5 at 128 chains, C3 at 1024 draws) between the product library and a variant, alternating runs
# in one call (boxes differ, clocks drift): tools/ab_gp_r4.sh <variant.so> [rounds]
V=$1; R=${2:-2}
for i in $(seq $R); do
  for lib in "" "$V"; do
    for cfg in "c5 --global-draws 128" "c3"; do
      EXOPLANET_AMD_LIB=$lib python bench.py --config $cfg --no-cpu-baseline --no-stats --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-product}'.split('/')[-1], '$cfg', round(d['ms_per_step'], 4), 'ms')"
    done
  done
done
