set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_all.txt 2>&1 || { tail -40 gpurun_out/r02_gpu_all.txt; exit 1; }
tail -2 gpurun_out/r02_gpu_all.txt
python bench.py 2>/dev/null | grep '^{' > gpurun_out/r02_bench_n1.json; cat gpurun_out/r02_bench_n1.json | cut -c1-400
python bench.py --gpus 1 --workload c5 --steps 20 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c5_full_one_gpu.json; cut -c1-300 gpurun_out/r02_bench_c5_full_one_gpu.json
python tools/report.py gpurun_out/r02_report_all_configs.md 2>/dev/null | tail -8
KB_TIMG=1 python tools/kbench.py c2 500 T 2>/dev/null | tail -2 > gpurun_out/r02_c2_transposed.txt
python tools/kbench.py c2 500 T 2>/dev/null | tail -2 >> gpurun_out/r02_c2_transposed.txt
cat gpurun_out/r02_c2_transposed.txt
