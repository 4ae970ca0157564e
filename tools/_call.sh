mkdir -p gpurun_out/r3d
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_sampling.py > gpurun_out/r3d/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r3d/pytest_gpu.txt
python -m pytest tests/test_gpu_sampling.py -x -q -m gpu > gpurun_out/r3d/pytest_gpu2.txt 2>&1
tail -5 gpurun_out/r3d/pytest_gpu2.txt
