#!/bin/bash
# headline steps, autograd form (default) against the one-call form: bash tools/bench_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "--config c2" "--config c4 --global-draws 64" "--config c4 --global-draws 512"; do
  for ag in 0 1; do
    ms=$(EXO_BENCH_ONE_CALL=$ag python $R/bench.py $cfg --no-cpu-baseline --no-extras --no-stats --steps 50 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$cfg one_call=$ag ms_per_step=$ms"
  done
done
