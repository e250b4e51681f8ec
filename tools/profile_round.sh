#!/bin/bash
# Evidence run for one round, on the GPU box:  bash tools/profile_round.sh r1
# Writes under gpurun_out/<tag>_*; tools/pmc_summary.py then turns the counter CSVs into
# gpurun_out/<tag>_kernel_counters.json (stamped with the hash of the kernel sources); that file
# and the kernel-stats CSV are copied to profiles/ by hand.
# rocprofv3 gets the interpreter itself after "--" (no env/bash hop: the profiler's preloaded
# library has already initialised the GPU), and the --pmc passes are separate from the trace pass.
set -eo pipefail
tag=${1:-r1}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench.json
cat $out/${tag}_bench.json
args="bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-breaking-dam"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o run -- python3 $args > $out/${tag}_stats.log 2>&1
pargs="bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-breaking-dam"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o run -- python3 $pargs > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o run -- python3 $pargs > $out/${tag}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/${tag}_pmc_valu -o run -- python3 $pargs > $out/${tag}_pmc_valu.log 2>&1
python3 tools/pmc_summary.py $tag
