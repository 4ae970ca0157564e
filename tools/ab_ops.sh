#!/bin/bash
# ops microbench over library variants: bash tools/ab_ops.sh <name|product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  lib=$R/tests/_build/variants/$v.so
  [ "$v" = product ] && lib=$R/exoplanet_amd/lib/libexoplanet_amd.so
  echo "$v $(EXOPLANET_AMD_LIB=$lib python $R/tools/ops_bench.py 2>&1 | tail -1)"
done
