set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_distributed.py -m gpu -x -q > gpurun_out/r02_dist_tests.txt 2>&1 || { tail -40 gpurun_out/r02_dist_tests.txt; exit 1; }
tail -2 gpurun_out/r02_dist_tests.txt
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > gpurun_out/r02_mfma_counters.txt || true
cat gpurun_out/r02_mfma_counters.txt
# rehearsal of the N = 2 bench on ONE GPU (gloo, both ranks on cuda:0, operators scaled to 4 %)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo --device 0 --scale 0.04 > gpurun_out/r02_bench_n2_rehearsal.json 2> gpurun_out/r02_bench_n2_rehearsal.err || { tail -30 gpurun_out/r02_bench_n2_rehearsal.err; exit 1; }
cat gpurun_out/r02_bench_n2_rehearsal.json
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err || { tail -30 gpurun_out/r02_bench_n1.err; exit 1; }
cat gpurun_out/r02_bench_n1.json
