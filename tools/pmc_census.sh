#!/bin/bash
# VALU wave-instructions of the two pair kernels by PHASE, from the hardware counter on ablated builds
# (tools/valu_census.py turns the differences into the census):
#   tools/build_variant.sh abl1 WORK -DSPH_ABLATE=1     density without SUM
#   tools/build_variant.sh abl2 WORK -DSPH_ABLATE=2     density without TEST / append / SUM
#   tools/build_variant.sh abl9 WORK -DSPH_ABLATE=9     density without the append (and so without SUM)
#   tools/build_variant.sh abl15 WORK -DSPH_ABLATE=15   density without the append and without SUM
#   tools/build_variant.sh abl21 WORK -DSPH_ABLATE=21   acceleration without its pair loops
#   bash tools/pmc_census.sh <tag>          (on the GPU box)  ->  gpurun_out/<tag>_census/<variant>/
# One rocprofv3 --pmc run per variant and pass; the program itself after "--"; tolerance-mode arithmetic.
set -eo pipefail
tag=${1:-r4}
out=gpurun_out/${tag}_census
mkdir -p $out
export TMPDIR=/tmp SPH_PMC_MODE=fast SPH_HIP_ALLOW_DIAGNOSTIC=1
passA="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
passB="SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"
unset SPH_HIP_LIBRARY
rocprofv3 --pmc $passA --output-format csv -d $out/base/A -o run -- python3 tools/pmc_one_step.py > $out/base_A.log 2>&1
rocprofv3 --pmc $passB --output-format csv -d $out/base/B -o run -- python3 tools/pmc_one_step.py > $out/base_B.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/base/C -o run -- python3 tools/pmc_one_step.py > $out/base_C.log 2>&1
for v in abl1 abl2 abl9 abl15 abl21; do
   export SPH_HIP_LIBRARY=$PWD/build/variants/$v.so
   rocprofv3 --pmc $passA --output-format csv -d $out/$v/A -o run -- python3 tools/pmc_one_step.py > $out/${v}_A.log 2>&1
done
unset SPH_HIP_LIBRARY
for v in base abl1 abl2 abl9 abl15 abl21; do
   echo "#### $v"; python3 tools/pmc_table.py $out/$v k_full_density_tiled k_full_accel_lists
done > $out/table.txt
find $out -name "*.csv" -size +2M -delete
cat $out/table.txt
