# rocprofv3 counter passes over one interleaved multi-RHS product: pmc_il.sh <case> <K>  -> gpurun_out/pmc_il_<case>/ ; table on stdout
# (counters in their own passes: --kernel-trace + --pmc only; FETCH_SIZE / WRITE_SIZE / the atomics one per pass)
set -e
R=$PWD
CASE=$1
K=$2
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_il_$CASE
rm -rf $O && mkdir -p $O
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "[pmc_il $CASE] pass $i: $line"
  timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --pmc $line -d $O/p$i -o p -- python3 $R/tools/mrhs_one.py $CASE $K 6 > /dev/null 2> $O/p$i.err || { grep -m2 -i "error\|exceeds" $O/p$i.err || true; }
done <<'LIST'
SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
FETCH_SIZE
WRITE_SIZE
TCC_EA0_ATOMIC_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_ATOMIC_sum TCC_READ_sum TCC_TAG_STALL_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
LIST
cd $R
python3 tools/pmc_table.py panel_kernel_il $O/p*
python3 tools/pmc_table.py il_pack_kernel $O/p1 | grep duration
python3 tools/pmc_table.py il_finish_kernel $O/p1 | grep duration
