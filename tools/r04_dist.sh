set -e
timeout -k 10 600 python -m pytest tests/test_gpu_multidevice.py -x -q 2>&1 | tail -3
BSM_DIST_FLAGS=0 timeout -k 10 600 python -m pytest tests/test_gpu_multidevice.py -x -q 2>&1 | tail -3
for i in 1 2; do
echo "-- flags (default on virtual devices)"; python tools/distbench.py c3 2>/dev/null
echo "-- events (BSM_DIST_FLAGS=0)"; BSM_DIST_FLAGS=0 python tools/distbench.py c3 2>/dev/null
done > gpurun_out/r04_distbench.txt
echo "-- flags c5s" >> gpurun_out/r04_distbench.txt; python tools/distbench.py c5s 2>/dev/null >> gpurun_out/r04_distbench.txt
echo "-- events c5s" >> gpurun_out/r04_distbench.txt; BSM_DIST_FLAGS=0 python tools/distbench.py c5s 2>/dev/null >> gpurun_out/r04_distbench.txt
cat gpurun_out/r04_distbench.txt
