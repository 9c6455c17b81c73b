# rocprofv3 passes on the tiled BEM fixture (tools/bem_real.py 400): kernel trace + atomic / traffic counters
set -e
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_bem
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/prof_bem/kt -o kt -- python3 $R/tools/bem_real.py 400 > $R/gpurun_out/prof_bem/run.txt 2> $R/gpurun_out/prof_bem/kt.err
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $R/gpurun_out/prof_bem/pmc_$c -o p -- python3 $R/tools/bem_real.py 400 > /dev/null 2> $R/gpurun_out/prof_bem/pmc_$c.err
done
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU -d $R/gpurun_out/prof_bem/pmc_sq -o p -- python3 $R/tools/bem_real.py 400 > /dev/null 2> $R/gpurun_out/prof_bem/pmc_sq.err
cd $R
python3 tools/kt_summary.py gpurun_out/prof_bem/kt gpurun_out/prof_bem/r02_bem_kernel_trace_by_grid.csv | head -8
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum sq; do
  python3 tools/pmc_summary.py mfma gpurun_out/prof_bem/r02_bem_$c.json "panel_kernel<bsm::c128, 8, true, true" gpurun_out/prof_bem/pmc_$c | cut -c1-400
done
cat gpurun_out/prof_bem/run.txt | tail -3
