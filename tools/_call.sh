mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_timed_config.py tests/test_gpu_ctypes_client.py -x -q -m gpu > gpurun_out/r3c/pytest_new.txt 2>&1
tail -30 gpurun_out/r3c/pytest_new.txt
python bench.py --config c4 --global-draws 64 --steps 20 --no-cpu-baseline --no-stats > gpurun_out/r3c/bench_c4.json 2> gpurun_out/r3c/bench_c4.err
python bench.py --config c5 --global-draws 128 --steps 10 --no-cpu-baseline --no-stats > gpurun_out/r3c/bench_c5.json 2> gpurun_out/r3c/bench_c5.err
tail -c 600 gpurun_out/r3c/bench_c4.json; tail -c 300 gpurun_out/r3c/bench_c4.err
tail -c 600 gpurun_out/r3c/bench_c5.json; tail -c 300 gpurun_out/r3c/bench_c5.err
