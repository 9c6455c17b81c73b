set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu --timeout 300 > gpurun_out/r04_mf_tests.txt 2>&1 || { grep -E "^E  |FAILED|passed|failed" gpurun_out/r04_mf_tests.txt | head -30; exit 1; }
tail -1 gpurun_out/r04_mf_tests.txt
python tools/multirhs.py bem_c64 bem_c128 2>/dev/null | cut -c1-200
python tools/multirhs.py bem_c64 bem_c128 2>/dev/null | cut -c1-200
