# alternating A/B of library variants on the GP configs: tools/ab_gp_alt.sh <reps> <variant|base> ...
R=$GRAFT_REPO_ROOT; reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
    a=$(python $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; print('%.3f'%json.loads(sys.stdin.read().strip().split('\n')[-1])['ms_per_step'])")
    b=$(python $R/bench.py --config c5 --global-draws 128 --steps 10 --warmup 2 --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; print('%.3f'%json.loads(sys.stdin.read().strip().split('\n')[-1])['ms_per_step'])")
    echo "$v c3_ms=$a c5_128_ms=$b"
  done
done
