# rocprofv3 counter passes over one tools/abbench.py case: pmc_case.sh <tag> <case>   (environment: the BSM_* knobs)
# -> gpurun_out/pmc_<tag>/p*/ ; summary by tools/pmc_table.py
set -e
R=$PWD
TAG=$1
CASE=$2
export TMPDIR=/tmp ABB_REPS=4
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "[pmc $TAG] pass $i: $line"
  timeout -k 5 150 rocprofv3 --output-format csv --kernel-trace --pmc $line -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python3 $R/tools/abbench.py $CASE > /dev/null 2> $R/gpurun_out/pmc_$TAG/p$i.err || { grep -m2 -i "error\|exceeds" $R/gpurun_out/pmc_$TAG/p$i.err || true; }
done <<'LIST'
TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_REQ_sum TCC_ATOMIC_sum TCC_READ_sum TCC_TAG_STALL_sum
TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
FETCH_SIZE
WRITE_SIZE
TCC_EA0_ATOMIC_sum
LIST
cd $R
python3 tools/pmc_table.py "_kernel<" gpurun_out/pmc_$TAG/p* > gpurun_out/pmc_$TAG.txt
rm -rf gpurun_out/pmc_$TAG
