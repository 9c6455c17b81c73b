#!/bin/bash
# Developer probe: SQ / LDS counters of one multi-RHS product (two --pmc passes), table to gpurun_out/mrhs_pmc_<case>_<K>.txt
# usage: tools/mrhs_pmc.sh <case> <K>
set -e
R=$PWD; O=$R/gpurun_out/mrhs_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/p1_$1_$2 -o p --output-format csv -- python3 $R/tools/mrhs_one.py $1 $2 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/p2_$1_$2 -o p --output-format csv -- python3 $R/tools/mrhs_one.py $1 $2 6 > /dev/null 2>&1
python3 $R/tools/pmc_table.py panel_kernel $O/p1_$1_$2 $O/p2_$1_$2 > $R/gpurun_out/mrhs_pmc_$1_$2.txt
find $O -name "*.csv" -delete
