#!/bin/bash
# C3 step time against the number of chunks of the time-parallel GP plan, sparse and dense mean: bash tools/c3_chunks.sh [chunks...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in "$@"; do
  for mode in 0 1; do
    ms=$(EXO_GP_CHUNKS=$c EXO_BENCH_DENSE_MEAN=$mode python $R/bench.py --config c3 --no-cpu-baseline --no-extras --no-stats --steps 20 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "chunks=$c dense_mean=$mode ms_per_step=$ms"
  done
done
