#!/bin/bash
# A/B build of the WHOLE library with extra flags on every translation unit:
#   tools/build_variant_all.sh <name> [extra hipcc flags ...]  ->  tests/_build/variants/<name>.so  (EXOPLANET_AMD_LIB=<path>)
# Per-file flags as in __graft_entry__.py (machine LICM off for the transit and celerite units).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
d=$R/tests/_build/variants/_obj/$name
mkdir -p $d
pids=()
for f in $R/exoplanet_amd/csrc/*.hip; do
  stem=$(basename ${f%.hip})
  extra=""
  case $stem in exo_transit|exo_celerite) extra="-mllvm -disable-machine-licm";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include $extra "$@" -c $f -o $d/$stem.o 2> $d/$stem.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tests/_build/variants/$name.so $d/*.o
echo built tests/_build/variants/$name.so
