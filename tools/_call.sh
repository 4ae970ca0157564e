mkdir -p gpurun_out/r3p
cd /tmp && export TMPDIR=/tmp
EXO_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3p/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-stats > $GRAFT_REPO_ROOT/gpurun_out/r3p/bench_forced.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3p/bench_forced.err
cd $GRAFT_REPO_ROOT
python tools/trace_overlap.py gpurun_out/r3p/trace gpurun_out/r3p/r03_forced_dist_overlap.txt
tail -c 300 gpurun_out/r3p/bench_forced.json
rm -rf gpurun_out/r3p/trace
EXO_BENCH_FORCE_DIST=1 python bench.py --steps 50 --no-extras --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('forced dist ms_per_step', d['ms_per_step'])"
python bench.py --steps 50 --no-extras --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('plain ms_per_step', d['ms_per_step'])"
