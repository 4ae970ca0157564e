#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> [extra hipcc flags for exo_transit.hip ...]
#   -> tests/_build/variants/<name>.so   (select with EXOPLANET_AMD_LIB=<path>; the other translation units come from
#   the product build's objects in exoplanet_amd/lib/_obj).  VARIANT_SRC=<file.hip>: another translation unit.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
src=${VARIANT_SRC:-exo_transit.hip}
stem=${src%.hip}
mkdir -p $R/tests/_build/variants/_obj
o=$R/tests/_build/variants/_obj/${name}_$stem.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include "$@" -c $R/exoplanet_amd/csrc/$src -o $o
others=$(ls $R/exoplanet_amd/lib/_obj/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tests/_build/variants/$name.so $o $others
echo built tests/_build/variants/$name.so
