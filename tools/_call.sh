bash tools/profile_r03.sh > gpurun_out/r03_log.txt 2>&1
tail -60 gpurun_out/r03_log.txt
