# rocprofv3 passes of round 3 (run on the GPU box from the repo root); everything lands in gpurun_out/prof3/ and
# is copied into profiles/ afterwards.  Counters are collected in their own passes (--kernel-trace + --pmc only).
set -e
R=$PWD
O=$R/gpurun_out/prof3
export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
cd /tmp
# 1. kernel trace + stats of the default bench command (the driver's command line)
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py > $O/r03_bench_c2_n1_under_rocprofv3.json 2> $O/bench_under_kt.err
# 2. HBM traffic of the C2 product: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --no-extra --steps 50 --warmup 5 > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $O/pmc_sq -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --no-extra --steps 50 --warmup 5 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum -d $O/pmc_tcc -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline --no-extra --steps 50 --warmup 5 > /dev/null 2> $O/pmc_tcc.err
# 3. MFMA counters on the C4 slice (128x128 fp32 blocks): expected 0
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 -d $O/pmc_mfma -o p -- python3 $R/tools/kbench.py c4s 20 > /dev/null 2> $O/pmc_mfma.err
# 4. the HBM-streaming legs: traffic, atomics, L2 hit rate, wave-state counters
for cfg in c2x20 c3 c4s c5s bem_c128; do
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/leg_${cfg}_fetch -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2> $O/leg_${cfg}.err
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/leg_${cfg}_write -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/leg_${cfg}.err
  rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -d $O/leg_${cfg}_tcc -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/leg_${cfg}.err
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -d $O/leg_${cfg}_sq -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/leg_${cfg}.err
done
# 5. the transposed product on the single image (C2): atomics per launch
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_ATOMIC_sum WRITE_SIZE -d $O/c2T_tcc -o p -- python3 $R/tools/kbench.py c2 100 T > /dev/null 2> $O/c2T.err
cd $R
python3 tools/pmc_summary.py traffic $O/r03_c2_pmc.json "panel_kernel<double" $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 54553920
python3 tools/pmc_summary.py mfma $O/r03_c4_mfma.json "panel_kernel<float" $O/pmc_mfma
{ echo "C2 product (bench.py --launch eager), rocprofv3 --pmc, per-dispatch means:"; python3 tools/pmc_table.py "panel_kernel<double, 8, true, false" $O/pmc_sq $O/pmc_tcc; } > $O/r03_c2_sq_tcc_counters.txt
{ for cfg in c2x20 c3 c4s c5s bem_c128; do echo "== $cfg (tools/abbench.py $cfg)"; python3 tools/pmc_table.py panel_kernel $O/leg_${cfg}_fetch $O/leg_${cfg}_write $O/leg_${cfg}_tcc $O/leg_${cfg}_sq; done; } > $O/r03_legs_counters.txt
{ echo "C2 transposed product on the single image (tools/kbench.py c2 100 T):"; python3 tools/pmc_table.py "panel_kernel" $O/c2T_tcc; python3 tools/pmc_table.py "scale_kernel" $O/c2T_tcc; } > $O/r03_c2_transposed_counters.txt
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/r03_c2_bench_default_kernel_stats.csv \;
python3 tools/kt_summary.py $O/kt $O/r03_c2_bench_default_kernel_trace_by_grid.csv > /dev/null
head -12 $O/r03_c2_bench_default_kernel_trace_by_grid.csv
# 6. un-profiled reference runs (bench.py reads the counter files of THIS build: copy them where it looks first)
cp $O/r03_c2_pmc.json $O/r03_c4_mfma.json $R/profiles/
python3 bench.py > $O/r03_bench_c2_n1.json 2> /dev/null
python3 tools/abbench.py > $O/r03_abbench.txt 2> /dev/null
python3 tools/kbench.py c2 500 T > $O/r03_c2_transposed.txt 2>&1 || true
KB_TIMG=1 python3 tools/kbench.py c2 500 T >> $O/r03_c2_transposed.txt 2>&1 || true
python3 tools/hostpath.py > $O/r03_hostpath.txt 2>&1 || true
for c in c3 c5s; do python3 tools/distbench.py $c; done > $O/r03_distbench.txt 2> /dev/null
python3 tools/multirhs.py > $O/r03_multirhs.txt 2> /dev/null
{ echo "nrhs = 1..9 right-hand sides in single products (tools/mrhs_sweep.py; batches of 8 / 4, padded remainders):"; python3 tools/mrhs_sweep.py c3 c5s c4s bem_f64 2> /dev/null; } > $O/r03_multirhs_sweep.txt
# 7. multi-RHS products: SQ / LDS counters (two --pmc passes each) and the timing-only ablations of the pipelined
#    kernel (experiment build)
{ echo "Multi right-hand-side products (bsm_mul_multi), rocprofv3 --kernel-trace --pmc (two passes), tools/mrhs_one.py, 6 dispatches each;"
  echo "SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES in quad-cycles summed over the chip, SQ_LDS_* in LDS cycles summed over the 256 CUs."
  for cfg in "c3 8" "c3 1" "c5s 8" "c4s 8"; do set -- $cfg; bash tools/mrhs_pmc.sh $1 $2; echo "== $1_$2"; cat gpurun_out/mrhs_pmc_$1_$2.txt; done; } > $O/r03_multirhs_counters.txt 2> /dev/null
{ echo "Timing-only ablations of the pipelined multi-RHS kernel (tools/ablate_multi.py, experiment build: results wrong by construction):"
  python3 tools/ablate_multi.py c3 8 2> /dev/null | tail -9; python3 tools/ablate_multi.py bem_f64 8 2> /dev/null | tail -9; } > $O/r03_multirhs_ablation.txt
find $O -name "*.csv" -size +2M -delete
rm -rf $O/kt $O/pmc_* $O/leg_*_fetch $O/leg_*_write $O/leg_*_tcc $O/leg_*_sq $O/c2T_tcc
ls -la $O
