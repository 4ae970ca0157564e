#!/bin/bash
# One GPU-box pass that produces everything profiles/ is built from:
#   gpurun_out/<tag>/bench.json            un-profiled bench.py line
#   gpurun_out/<tag>/trace/                rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>/pmc_fetch|pmc_write/  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes)
#   gpurun_out/<tag>/configs.json          tools/bench_configs.py
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>$out/bench.err | tail -1 > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-graph > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-graph > /dev/null 2>&1
python $R/tools/bench_configs.py > $out/configs.json 2>$out/configs.err
find $out -name "*.csv" | head -20
cat $out/bench.json | cut -c1-300
