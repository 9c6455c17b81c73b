# rocprofv3 passes of round 5 (run on the GPU box from the repo root); everything lands in gpurun_out/prof5/ and is copied into
# profiles/ afterwards.  Counters are collected in their own passes (--kernel-trace + --pmc only).  BSM_LIB_BEFORE = a build of the
# round's starting point (tools/build_prev.sh) for the before / after counters of the BEM legs; skipped when absent.
set -e
R=$PWD
O=$R/gpurun_out/prof5
export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
BEFORE=${BSM_LIB_BEFORE:-$R/blocksparsematrices.jl_amd/libbsmrocm_head.so}
cd /tmp
say() { echo "[profile_round] $*"; }
# 1. kernel trace + stats of the default bench command (the driver's command line)
say "kernel trace of the default bench"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-live-pmc > $O/r05_bench_c2_n1_under_rocprofv3.json 2> $O/bench_under_kt.err
# 2. HBM traffic of the C2 product: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  say "C2 $c"
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python3 $R/bench.py --pmc-child > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $O/pmc_sq -o p -- python3 $R/bench.py --pmc-child > /dev/null 2> $O/pmc_sq.err
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum -d $O/pmc_tcc -o p -- python3 $R/bench.py --pmc-child > /dev/null 2> $O/pmc_tcc.err
# 3. MFMA counters on the C4 slice (128x128 fp32 blocks): expected 0
say "MFMA counters, C4 slice"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 -d $O/pmc_mfma -o p -- python3 $R/tools/kbench.py c4s 20 > /dev/null 2> $O/pmc_mfma.err
# 3b. ... and on the products that DO run on the matrix pipe: 8 complex right-hand sides on the BEM fixture
# (one memory counter per pass: FETCH_SIZE / WRITE_SIZE / the atomics do not fit one configuration)
for cfg in bem_c128 bem_c64; do
  say "MFMA counters, $cfg x 8"
  timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/mfma_${cfg} -o p -- python3 $R/tools/mrhs_one.py $cfg 8 10 > /dev/null 2> $O/mfma_${cfg}.err
  for c in TCC_EA0_ATOMIC_sum WRITE_SIZE FETCH_SIZE; do
    say "  $cfg x 8 $c"
    timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/mfma_${cfg}_$c -o p -- python3 $R/tools/mrhs_one.py $cfg 8 10 > /dev/null 2>> $O/mfma_${cfg}.err
  done
done
# 4. the HBM-streaming legs: traffic, atomics, L2 hit rate, wave-state counters
export ABB_REPS=8
for cfg in c2x20 c3 c4s c5s bem_c128 bem_f64 bem_c64 bem_f32; do
  say "leg $cfg"
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/leg_${cfg}_fetch -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2> $O/leg_${cfg}.err
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/leg_${cfg}_write -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/leg_${cfg}.err
  rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -d $O/leg_${cfg}_tcc -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/leg_${cfg}.err
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -d $O/leg_${cfg}_sq -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/leg_${cfg}.err
done
# 4b. the BEM legs BEFORE this round's kernel changes (atomics and write traffic)
if [ -f "$BEFORE" ]; then
  for cfg in bem_c128 bem_f64 bem_c64 bem_f32; do
    say "leg $cfg with the round's starting library"
    BSM_LIB=$BEFORE rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/before_${cfg}_write -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2> $O/before_${cfg}.err
    BSM_LIB=$BEFORE rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -d $O/before_${cfg}_tcc -o p -- python3 $R/tools/abbench.py $cfg > /dev/null 2>> $O/before_${cfg}.err
  done
fi
unset ABB_REPS
# 5. the transposed product on the single image (C2): atomics per launch
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_ATOMIC_sum WRITE_SIZE -d $O/c2T_tcc -o p -- python3 $R/tools/kbench.py c2 100 T > /dev/null 2> $O/c2T.err
cd $R
python3 tools/pmc_summary.py traffic $O/r05_c2_pmc.json "panel_kernel<double" $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 54553920
python3 tools/pmc_summary.py mfma $O/r05_c4_mfma.json "panel_kernel<float" $O/pmc_mfma
{ echo "C2 product (bench.py --pmc-child: 60 eager launches), rocprofv3 --pmc, per-dispatch means:"; python3 tools/pmc_table.py "panel_kernel<double, 8, true, false" $O/pmc_sq $O/pmc_tcc; } > $O/r05_c2_sq_tcc_counters.txt
{ for cfg in c2x20 c3 c4s c5s bem_c128 bem_f64 bem_c64 bem_f32; do echo "== $cfg (tools/abbench.py $cfg)"; python3 tools/pmc_table.py panel_kernel $O/leg_${cfg}_fetch $O/leg_${cfg}_write $O/leg_${cfg}_tcc $O/leg_${cfg}_sq; done
  if [ -f "$BEFORE" ]; then for cfg in bem_c128 bem_f64 bem_c64 bem_f32; do echo "== $cfg BEFORE (the library the round started from)"; python3 tools/pmc_table.py panel_kernel $O/before_${cfg}_write $O/before_${cfg}_tcc; done; fi; } > $O/r05_legs_counters.txt
{ for cfg in bem_c128 bem_c64; do echo "== $cfg x 8 (tools/mrhs_one.py $cfg 8 10): the interleaved 8-column pass (panel_kernel_il) on the matrix pipe, per-dispatch means"; python3 tools/pmc_table.py panel_kernel_il $O/mfma_${cfg} $O/mfma_${cfg}_TCC_EA0_ATOMIC_sum $O/mfma_${cfg}_WRITE_SIZE $O/mfma_${cfg}_FETCH_SIZE; done; } > $O/r05_multirhs_mfma_counters.txt
{ echo "C2 transposed product on the single image (tools/kbench.py c2 100 T):"; python3 tools/pmc_table.py "panel_kernel" $O/c2T_tcc; python3 tools/pmc_table.py "scale_kernel" $O/c2T_tcc; } > $O/r05_c2_transposed_counters.txt
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/r05_c2_bench_default_kernel_stats.csv \;
python3 tools/kt_summary.py $O/kt $O/r05_c2_bench_default_kernel_trace_by_grid.csv > /dev/null
head -12 $O/r05_c2_bench_default_kernel_trace_by_grid.csv
# 6. un-profiled reference runs (bench.py reads the counter files of THIS build: copy them where it looks first)
cp $O/r05_c2_pmc.json $O/r05_c4_mfma.json $R/profiles/
say "bench.py --gpus 1 --workload c5 (the anchor of the N > 1 lines)"
python3 bench.py --gpus 1 --workload c5 --steps 20 --warmup 3 > $O/r05_bench_c5_full_one_gpu.json 2> /dev/null
python3 - $O/r05_bench_c5_full_one_gpu.json $O/r05_c5_n1.json <<'PY'
import json, sys
sys.path.insert(0, ".")
from bsm_amd import _lib
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
out = {"value": d["value"], "unit": "GB/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "exchange_us": d["config"]["exchange_us"],
       "local_kernel_us_max": d["config"]["local_kernel_us_max"], "parity_relerr": d["config"]["parity_relerr"],
       "workload": "C5 (SymmetricBlockMatrix 5M x 5M, 16-256 blocks, fp64, 29.0 GB algorithmic) on ONE MI355X through the N > 1 code path: "
                   "python bench.py --gpus 1 --workload c5 --steps 20 --warmup 3",
       "build": _lib.lib().bsm_version().decode().split("build ")[-1]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
cp $O/r05_c5_n1.json $R/profiles/
say "default bench"
python3 bench.py > $O/r05_bench_c2_n1.json 2> /dev/null
python3 tools/abbench.py > $O/r05_abbench.txt 2> /dev/null
python3 tools/kbench.py c2 500 T > $O/r05_c2_transposed.txt 2>&1 || true
KB_TIMG=1 python3 tools/kbench.py c2 500 T >> $O/r05_c2_transposed.txt 2>&1 || true
python3 tools/multirhs.py > $O/r05_multirhs.txt 2> /dev/null || true
say "multi-device fan-out on virtual devices"
{ for r in 1 2; do echo "-- default (parts on the caller's device share its stream: no ordering packets; work vectors kept zero by the finish kernels)"; python3 tools/distbench.py c3 2> /dev/null
  echo "-- BSM_DIST_REZERO=0 (a w = 0 launch in front of every part's product, as in round 3)"; BSM_DIST_REZERO=0 python3 tools/distbench.py c3 2> /dev/null
  echo "-- BSM_DIST_ONE_STREAM=0: every part on a stream of its own, ordered by flags (what parts on DISTINCT devices pay)"; BSM_DIST_ONE_STREAM=0 python3 tools/distbench.py c3 2> /dev/null
  echo "-- BSM_DIST_ONE_STREAM=0 BSM_DIST_FLAGS=0: ... ordered by events"; BSM_DIST_ONE_STREAM=0 BSM_DIST_FLAGS=0 python3 tools/distbench.py c3 2> /dev/null; done; } > $O/r05_distbench.txt
say "ablation of the matrix-pipe multi-RHS kernels"
{ python3 tools/ablate_multi.py bem_c128 8 2> /dev/null | tail -14; python3 tools/ablate_multi.py bem_c64 8 2> /dev/null | tail -14; } > $O/r05_multirhs_mfma_ablation.txt
say "ablations of the fused kernel on the tiled BEM fixture (experiment build)"
{ for t in c128 f64 c64 f32; do python3 tools/ablate.py 400 $t 3 2> /dev/null; done
  echo; echo "(tools/ablate.py on the experiment build, make -C blocksparsematrices.jl_amd/csrc exp: every variant drops one part of the fused kernel -- results are wrong by construction, only the times mean something; interleaved rounds in one process)"; } > $O/r05_bem_ablation.txt
say "interleaved multi-RHS pass: counters, ablations, wave timeline"
{ for c in "bem_c128 8" "bem_c64 8" "bem_f64 8" "bem_f64 16" "c3 16"; do set -- $c; echo "== $1 x $2 (tools/pmc_il.sh $1 $2: tools/mrhs_one.py under rocprofv3 --pmc, one pass per line of counters), per launch"; bash tools/pmc_il.sh $1 $2 2> /dev/null | grep "mean"; done; } > $O/r05_il_counters.txt
{ python3 tools/ablate_multi.py bem_c128 8 2> /dev/null | tail -14; python3 tools/ablate_multi.py bem_c64 8 2> /dev/null | tail -14; python3 tools/ablate_multi.py bem_f64 8 2> /dev/null | tail -14; python3 tools/ablate_multi.py bem_f64 16 2> /dev/null | tail -14
  echo; echo "(tools/ablate_multi.py on the experiment build; the times include the pack and finish passes of the interleaved pass: 24 + 30 us ComplexF64, see r05_il_counters.txt)"; } > $O/r05_il_ablation.txt
{ BSM_LIB=$R/blocksparsematrices.jl_amd/libbsmrocm_trace.so python3 tools/il_trace.py bem_c128 8 2> /dev/null | tail -10; BSM_LIB=$R/blocksparsematrices.jl_amd/libbsmrocm_trace.so python3 tools/il_trace.py bem_c64 8 2> /dev/null | tail -10; } > $O/r05_il_wavetrace.txt
{ echo "BSM_MULTI_IL=0 (the round-4 kernels):"; BSM_MULTI_IL=0 python3 tools/multirhs.py bem_c128 bem_c64 bem_f64 bem_f32 c3 c3_f32 c5s 2> /dev/null | grep rhs; echo "default (interleaved pass):"; python3 tools/multirhs.py bem_c128 bem_c64 bem_f64 bem_f32 c3 c3_f32 c5s 2> /dev/null | grep rhs; } > $O/r05_il_ab.txt
say "one-rank RCCL loopback of the N > 1 step"
{ for a in "0 1" "8 1"; do python3 tools/loopback_trace.py 0.125 nccl $a 2>&1 | grep "rows\|compute\|first"; done; } > $O/r05_loopback_first_steps.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/lb -o lb -- python3 $R/tools/loopback_trace.py 0.125 nccl 8 > /dev/null 2>&1 || true
cd $R
python3 tools/kt_timeline.py $O/lb 26 > $O/r05_loopback_timeline.txt 2> /dev/null || true
python3 bench.py --gpus 1 --workload c5 --loopback --steps 20 --warmup 3 --no-extra 2> /dev/null | grep "^{" > $O/r05_bench_c5_loopback_rccl.json || true
python3 bench.py --gpus 1 --workload c5 --loopback --steps 20 --warmup 3 --no-extra --scale 0.125 2> /dev/null | grep "^{" > $O/r05_bench_c5_loopback_rccl_eighth.json || true
python3 tools/report.py $O/r05_report_all_configs.md > /dev/null 2>&1 || true
find $O -name "*.csv" -size +2M -delete
rm -rf $O/kt $O/pmc_* $O/leg_* $O/before_* $O/c2T_tcc $O/mfma_bem_* $O/lb
ls -la $O
