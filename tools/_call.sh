mkdir -p gpurun_out/r3k
python -m pytest tests -x -q -m gpu > gpurun_out/r3k/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r3k/pytest_gpu.txt
python bench.py > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err
tail -c 400 gpurun_out/r3k/bench.err
