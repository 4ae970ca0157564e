#!/bin/bash
# A/B of one environment switch on one box, alternating: bash tools/ab_env.sh <rounds> <VAR> <value> <value> ... -> configs_ms per run
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; var=$2; shift 2
for i in $(seq $rounds); do
  for v in "$@"; do
    line=$(env $var=$v python $R/bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read())['configs_ms']))")
    echo "$var=$v $line"
  done
done
