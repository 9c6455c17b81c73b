# rocprofv3 counter passes over one multi-RHS product: pmc_multi.sh <case> <K>   -> gpurun_out/pmc_multi_<case>/ ; table on stdout
set -e
R=$PWD
CASE=$1
K=$2
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_multi_$CASE
rm -rf $O && mkdir -p $O
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "[pmc_multi $CASE] pass $i: $line"
  timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --pmc $line -d $O/p$i -o p -- python3 $R/tools/mrhs_one.py $CASE $K 6 > /dev/null 2> $O/p$i.err || { grep -m2 -i "error\|exceeds" $O/p$i.err || true; }
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_REQ_sum TCC_ATOMIC_sum TCC_READ_sum TCC_TAG_STALL_sum
LIST
cd $R
python3 tools/pmc_table.py panel_kernel_multi $O/p*
