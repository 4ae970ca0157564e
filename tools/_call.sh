mkdir -p gpurun_out/r3q
R=$GRAFT_REPO_ROOT
for r in 1 2; do
for v in base gp_span2w2 gp_span2w3; do for ch in 0 384 192; do
    if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
    a=$(EXO_GP_CHUNKS=$ch python $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-stats 2>/dev/null | python -c "import json,sys; print('%.3f'%json.loads(sys.stdin.read().strip().split('\n')[-1])['ms_per_step'])")
    echo "$v chunks=$ch c3_ms=$a"
done; done; done > gpurun_out/r3q/ab.txt 2>&1
cat gpurun_out/r3q/ab.txt
