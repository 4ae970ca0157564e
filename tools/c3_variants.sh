#!/bin/bash
# C3 step time for A/B builds of the library (tests/_build/variants/<name>.so; "base" = the product build), sparse and dense mean
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
  for mode in 0 1; do
    ms=$(EXO_GP_CHUNKS=${CHUNKS:-0} EXO_BENCH_DENSE_MEAN=$mode python $R/bench.py --config ${CFG:-c3} ${BARGS} --no-cpu-baseline --no-extras --no-stats --steps 20 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "variant=$v dense_mean=$mode ms_per_step=$ms"
  done
done
