# rocprofv3 passes of round 2 (run on the GPU box from the repo root); summaries land in gpurun_out/
set -e
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
# 1. kernel trace + stats of the default bench command
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt -- python3 $R/bench.py > $R/gpurun_out/prof/bench_under_kt.json 2> $R/gpurun_out/prof/bench_under_kt.err
# 2. PMC passes (one counter per pass, kernel trace only) on the C2 product
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $R/gpurun_out/prof/pmc_$c -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --no-extra --steps 50 --warmup 5 > /dev/null 2> $R/gpurun_out/prof/pmc_$c.err
done
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/prof/pmc_sq -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --no-extra --steps 50 --warmup 5 > /dev/null 2> $R/gpurun_out/prof/pmc_sq.err
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof/pmc_tcc -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --no-extra --steps 50 --warmup 5 > /dev/null 2> $R/gpurun_out/prof/pmc_tcc.err
# 3. MFMA counters on the C4 slice (128x128 fp32 blocks): expected 0
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 -d $R/gpurun_out/prof/pmc_mfma -o p -- python3 $R/tools/kbench.py c4s 20 > /dev/null 2> $R/gpurun_out/prof/pmc_mfma.err
cd $R
python3 tools/pmc_summary.py traffic gpurun_out/prof/r02_c2_pmc.json "panel_kernel<double" gpurun_out/prof/pmc_FETCH_SIZE gpurun_out/prof/pmc_WRITE_SIZE 54553920
python3 tools/pmc_summary.py mfma gpurun_out/prof/r02_c4_mfma.json "panel_kernel<float" gpurun_out/prof/pmc_mfma
python3 tools/pmc_summary.py mfma gpurun_out/prof/r02_c2_sq.json "panel_kernel<double" gpurun_out/prof/pmc_sq
python3 tools/pmc_summary.py mfma gpurun_out/prof/r02_c2_tcc.json "panel_kernel<double" gpurun_out/prof/pmc_tcc
find gpurun_out/prof/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof/r02_c2_bench_default_kernel_stats.csv \;
cat gpurun_out/prof/r02_c2_bench_default_kernel_stats.csv | head -12
cat gpurun_out/prof/bench_under_kt.json; python tools/hostpath.py > gpurun_out/r02_hostpath.txt 2>&1; cat gpurun_out/r02_hostpath.txt
