for i in 1 2; do for lib in "" tests/_build/variants/nofuse.so tests/_build/variants/fuse64.so; do for cfg in "c5 --global-draws 128" "c3"; do
EXOPLANET_AMD_LIB=$lib python bench.py --config $cfg --no-cpu-baseline --no-stats --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-product}', '$cfg', round(d['ms_per_step'],4))"
done; done; done
