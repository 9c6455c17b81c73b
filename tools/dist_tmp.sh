set -e
timeout -k 10 600 python -m pytest tests/test_gpu_multidevice.py tests/test_c_abi_from_c.py -x -q -m gpu --timeout 120 > gpurun_out/r04_dist_tests.txt 2>&1 || { tail -40 gpurun_out/r04_dist_tests.txt; exit 1; }
tail -2 gpurun_out/r04_dist_tests.txt
{ for r in 1 2; do echo "-- default (one stream for parts on the caller's device, work vectors kept zero)"; python tools/distbench.py c3 2>/dev/null
echo "-- BSM_DIST_REZERO=0"; BSM_DIST_REZERO=0 python tools/distbench.py c3 2>/dev/null
echo "-- BSM_DIST_ONE_STREAM=0 (flags)"; BSM_DIST_ONE_STREAM=0 python tools/distbench.py c3 2>/dev/null; 
echo "-- BSM_DIST_ONE_STREAM=0 BSM_DIST_FLAGS=0 (events)"; BSM_DIST_ONE_STREAM=0 BSM_DIST_FLAGS=0 python tools/distbench.py c3 2>/dev/null; done; } > gpurun_out/r04_distbench2.txt
cat gpurun_out/r04_distbench2.txt
