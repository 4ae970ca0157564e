#!/bin/bash
# A/B of library builds on the GP-heavy legs only (tools/ab_gp.py), alternating: bash tools/ab_gp_variants.sh <rounds> <name|product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for i in $(seq $rounds); do
  for v in "$@"; do
    lib=$R/tests/_build/variants/$v.so
    [ "$v" = product ] && lib=$R/exoplanet_amd/lib/libexoplanet_amd.so
    EXOPLANET_AMD_LIB=$lib python $R/tools/ab_gp.py 2>/dev/null | tail -1
  done
done
