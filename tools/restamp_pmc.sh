# The two stamped counter files bench.py reads (profiles/r05_c2_pmc.json, r05_c4_mfma.json) for THIS build of the kernels: the
# three rocprofv3 passes of steps 2-3 of tools/profile_round5.sh alone (run on the GPU box from the repo root, ~1 minute).
set -e
R=$PWD
O=$R/gpurun_out/restamp
export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python3 $R/bench.py --pmc-child > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 -d $O/pmc_mfma -o p -- python3 $R/tools/kbench.py c4s 20 > /dev/null 2> $O/pmc_mfma.err
cd $R
python3 tools/pmc_summary.py traffic $O/r05_c2_pmc.json "panel_kernel<double" $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 54553920
python3 tools/pmc_summary.py mfma $O/r05_c4_mfma.json "panel_kernel<float" $O/pmc_mfma
cp $O/r05_c2_pmc.json $O/r05_c4_mfma.json $R/profiles/
cp $O/r05_c2_pmc.json $O/r05_c4_mfma.json $R/gpurun_out/
cat $O/r05_c2_pmc.json
