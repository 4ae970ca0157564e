#!/bin/bash
# A/B of library builds on one box, alternating: bash tools/ab_variants.sh <rounds> <name|product> ...
#   (names = tests/_build/variants/<name>.so; "product" = exoplanet_amd/lib/libexoplanet_amd.so) -> configs_ms per run
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for i in $(seq $rounds); do
  for v in "$@"; do
    lib=$R/tests/_build/variants/$v.so
    [ "$v" = product ] && lib=$R/exoplanet_amd/lib/libexoplanet_amd.so
    line=$(EXOPLANET_AMD_LIB=$lib python $R/bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read())['configs_ms']))")
    echo "$v $line"
  done
done
