set -e
timeout -k 10 600 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_distributed.py -x -q --timeout 120 > gpurun_out/r04_dist2_tests.txt 2>&1 || { tail -30 gpurun_out/r04_dist2_tests.txt; exit 1; }; tail -3 gpurun_out/r04_dist2_tests.txt
for i in 1 2; do
echo "-- flags (default on virtual devices)"; python tools/distbench.py c3 2>/dev/null
echo "-- events (BSM_DIST_FLAGS=0)"; BSM_DIST_FLAGS=0 python tools/distbench.py c3 2>/dev/null
done > gpurun_out/r04_distbench.txt
cat gpurun_out/r04_distbench.txt
python bench.py --gpus 2 --backend gloo --device 0 --scale 0.04 --steps 10 --warmup 2 > gpurun_out/r04_bench_n2_rehearsal.json 2> gpurun_out/r04_bench_n2_rehearsal.err || tail -20 gpurun_out/r04_bench_n2_rehearsal.err
cut -c1-1500 gpurun_out/r04_bench_n2_rehearsal.json
