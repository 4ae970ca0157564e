for i in 1 2; do for lib in "" tests/_build/variants/gfence.so; do
EXOPLANET_AMD_LIB=$lib python bench.py --config c5 --global-draws 128 --no-cpu-baseline --no-stats --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-product}', d['ms_per_step'])"
done; done
