#!/bin/bash
# Counter passes over one 4M-particle dam-break step (tools/pmc_one_step.py), one rocprofv3 run
# per pass (SQ has 8 slots, TCC 4): bash tools/pmc_passes.sh <tag>  ->  gpurun_out/pmc_<tag>/pass*/
# then  python3 tools/pmc_table.py gpurun_out/pmc_<tag> k_full
set -eo pipefail
tag=${1:-x}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/pass$n -o run -- python3 tools/pmc_one_step.py > $out/pass$n.log 2>&1; }
run 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run 2 SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR
run 3 SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH
run 4 SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_VMEM
run 5 GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_WRITE TCP_TOTAL_READ
run 6 TCC_REQ TCC_WRITE TCC_READ TCC_HIT
python3 tools/pmc_table.py $out k_full > $out/table.txt
cat $out/table.txt
