set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_distributed.py -m gpu -x -q > gpurun_out/r02_dist_tests.txt 2>&1 || { tail -40 gpurun_out/r02_dist_tests.txt; exit 1; }
tail -2 gpurun_out/r02_dist_tests.txt
# rehearsal of the N = 2 / N = 3 bench on ONE GPU (gloo, every rank on cuda:0, operators scaled to 4 %)
for n in 2 3; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n \
   bench.py --gpus $n --steps 20 --warmup 3 --backend gloo --device 0 --scale 0.04 2> gpurun_out/r02_bench_n${n}_rehearsal.err | grep '^{' > gpurun_out/r02_bench_n${n}_rehearsal.json || { tail -30 gpurun_out/r02_bench_n${n}_rehearsal.err; exit 1; }
cat gpurun_out/r02_bench_n${n}_rehearsal.json
done
python bench.py --gpus 1 --workload c5 --steps 20 --warmup 3 --no-extra 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c5_full_one_gpu.json
cat gpurun_out/r02_bench_c5_full_one_gpu.json
