#!/bin/bash
# instruction counts of two library builds, same box: bash tools/pmc_two.sh libA.so libB.so
set -eo pipefail
export TMPDIR=/tmp
for so in "$@"; do
   tag=$(basename $so .so)
   out=gpurun_out/pmc2_$tag
   mkdir -p $out
   export SPH_HIP_LIBRARY=$PWD/$so
   rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/p1 -o run -- python3 tools/pmc_one_step.py > $out/p1.log 2>&1
   rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU --output-format csv -d $out/p2 -o run -- python3 tools/pmc_one_step.py > $out/p2.log 2>&1
   echo "#### $tag"; python3 tools/pmc_table.py $out k_full_density
done
